// The reference's main() (cuda/nmf.cu:30-51) asking the library for several GPUs: the call is still one update_div_ex on
// host matrices -- column shards, threads, RCCL and the gather of H are the library's business (SURVEY 8b "Threading").
//   sharded_main <dir> <n_devices> [emulate]     emulate: n ranks on ONE device (one-GPU boxes; see nmf_opts.emulate_shards)
// reads <dir>/X.bin W.bin H.bin, runs MAX_ITER iterations, writes <dir>/Wout.bin Hout.bin, prints one summary line.
// Build:  g++ -O2 -Iinclude examples/sharded_main.cpp -Lnmf-gpu_amd -lnmf_mi355x -Wl,-rpath,$PWD/nmf-gpu_amd -o sharded_main
#include "nmf_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define MAX_ITER 200               // cuda/nmf.cu:10

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: %s <dir> <n_devices> [emulate]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const int n = atoi(argv[2]);
    const bool emulate = argc > 3 && !strcmp(argv[3], "emulate");
    matrix X, H, W;
    if (nmf_read_matrix(&X, (dir + "/X.bin").c_str()) || nmf_read_matrix(&H, (dir + "/H.bin").c_str()) || nmf_read_matrix(&W, (dir + "/W.bin").c_str())) {
        fprintf(stderr, "%s\n", nmf_last_error());
        return 1;
    }
    nmf_opts o;
    nmf_default_opts(&o);
    o.max_iter = MAX_ITER;
    if (emulate) o.emulate_shards = n; else o.n_devices = n;
    nmf_result r;
    const int st = update_div_ex(W, H, X, &o, &r);
    if (st != NMF_OK) { fprintf(stderr, "update_div_ex: %s (%s)\n", nmf_status_string(st), nmf_last_error()); return st; }
    printf("shards %d iterations %d w_replicas_identical %d total_s %.4f\n", r.n_shards, r.iterations, r.w_replicas_identical, r.t[NMF_T_TOTAL]);
    if (nmf_write_matrix(W, (dir + "/Wout.bin").c_str()) || nmf_write_matrix(H, (dir + "/Hout.bin").c_str())) return 1;
    nmf_destroy_matrix(&X); nmf_destroy_matrix(&H); nmf_destroy_matrix(&W);
    return 0;
}
