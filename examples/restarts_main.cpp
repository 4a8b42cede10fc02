// Multi-restart NMF over the GPUs of a node ("replicas only", paper section 3.2: "multiple random initializations and
// choose the best"): one update_div_restarts call on host matrices; the library deals restart i to worker i % n_workers,
// each worker a host thread with its own batched solver and its own copy of X -- no communicator, no collective.
//   restarts_main <dir> <n_restarts> <n_workers> [<device> ...]
// reads <dir>/X.bin and <dir>/W<i>.bin, <dir>/H<i>.bin (i = 0 .. n_restarts-1), runs MAX_ITER iterations of each, writes
// <dir>/Wout<i>.bin, Hout<i>.bin, prints "best <i>" and every restart's final KL divergence.
// Devices: the optional list names one HIP ordinal per worker (a device may appear twice: two workers on one GPU -- what a
// one-GPU box can run of this); without it the workers use devices 0 .. n_workers-1.
// Build:  g++ -O2 -Iinclude examples/restarts_main.cpp -Lnmf-gpu_amd -lnmf_mi355x -Wl,-rpath,$PWD/nmf-gpu_amd -o restarts_main
#include "nmf_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define MAX_ITER 200               // cuda/nmf.cu:10

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s <dir> <n_restarts> <n_workers> [<device> ...]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    const int R = atoi(argv[2]), G = atoi(argv[3]);
    if (R < 1 || G < 1) return 2;
    std::vector<int> devices;
    for (int i = 4; i < argc; ++i) devices.push_back(atoi(argv[i]));
    if (!devices.empty() && (int)devices.size() != G) { fprintf(stderr, "need one device per worker\n"); return 2; }
    matrix X;
    if (nmf_read_matrix(&X, (dir + "/X.bin").c_str())) { fprintf(stderr, "%s\n", nmf_last_error()); return 1; }
    std::vector<matrix> W((size_t)R), H((size_t)R);
    for (int i = 0; i < R; ++i)
        if (nmf_read_matrix(&W[(size_t)i], (dir + "/W" + std::to_string(i) + ".bin").c_str()) || nmf_read_matrix(&H[(size_t)i], (dir + "/H" + std::to_string(i) + ".bin").c_str())) {
            fprintf(stderr, "%s\n", nmf_last_error());
            return 1;
        }
    nmf_opts o;
    nmf_default_opts(&o);
    o.max_iter = MAX_ITER;
    o.n_devices = G;
    o.devices = devices.empty() ? nullptr : devices.data();
    int best = -1;
    std::vector<double> kl((size_t)R);
    const int st = update_div_restarts(W.data(), H.data(), R, X, &o, &best, kl.data());
    if (st != NMF_OK) { fprintf(stderr, "update_div_restarts: %s (%s)\n", nmf_status_string(st), nmf_last_error()); return st; }
    printf("best %d\n", best);
    for (int i = 0; i < R; ++i) {
        printf("restart %d kl %.17g\n", i, kl[(size_t)i]);
        if (nmf_write_matrix(W[(size_t)i], (dir + "/Wout" + std::to_string(i) + ".bin").c_str()) || nmf_write_matrix(H[(size_t)i], (dir + "/Hout" + std::to_string(i) + ".bin").c_str())) return 1;
        nmf_destroy_matrix(&W[(size_t)i]); nmf_destroy_matrix(&H[(size_t)i]);
    }
    nmf_destroy_matrix(&X);
    return 0;
}
