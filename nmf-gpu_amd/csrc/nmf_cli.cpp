// nmf_cli.cpp -- command-line driver, the counterpart of the reference's main()
// (cuda/nmf.cu:30-51): read X, H, W (.bin), run update_div, write Wout, Hout.
// The reference hard-codes ../X.bin ../H.bin ../W.bin -> ../Wout.bin ../Hout.bin
// (cuda/nmf.cu:37-45) and compile-time MAX_ITER / CONVERGE_THRESH (cuda/nmf.cu:9-11);
// those stay the defaults, every one of them is a flag here.
#include "../../include/nmf_mi355x.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [--X f] [--W f] [--H f] [--Wout f] [--Hout f] [--iters n] [--thresh x]\n"
            "          [--check n] [--verbose] [--timers] [--path auto|fused|unfused] [--device d]\n"
            "defaults follow cuda/nmf.cu:9-11,37-45: ../X.bin ../W.bin ../H.bin -> ../Wout.bin ../Hout.bin,\n"
            "200 iterations, threshold 0, check every 25.\n", argv0);
}

int main(int argc, char **argv) {
    std::string fx = "../X.bin", fw = "../W.bin", fh = "../H.bin", fwo = "../Wout.bin", fho = "../Hout.bin";
    nmf_opts o;
    nmf_default_opts(&o);
    bool timers = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(argv[0]); exit(2); } return argv[++i]; };
        if (a == "--X") fx = next();
        else if (a == "--W") fw = next();
        else if (a == "--H") fh = next();
        else if (a == "--Wout") fwo = next();
        else if (a == "--Hout") fho = next();
        else if (a == "--iters") o.max_iter = atoi(next());
        else if (a == "--thresh") o.converge_thresh = (float)atof(next());
        else if (a == "--check") o.iter_check = atoi(next());
        else if (a == "--verbose") o.verbose = 1;
        else if (a == "--timers") timers = true;
        else if (a == "--device") o.device = atoi(next());
        else if (a == "--path") {
            const std::string p = next();
            o.path = p == "fused" ? NMF_PATH_FUSED : p == "unfused" ? NMF_PATH_UNFUSED : NMF_PATH_AUTO;
        } else { usage(argv[0]); return a == "--help" || a == "-h" ? 0 : 2; }
    }
    if (timers) o.use_graph = 0;
    matrix X, W, H;
    int st;
    if ((st = nmf_read_matrix(&X, fx.c_str())) || (st = nmf_read_matrix(&H, fh.c_str())) || (st = nmf_read_matrix(&W, fw.c_str()))) {
        fprintf(stderr, "nmf: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        return st;
    }
    printf("read %s [%ix%i]\nread %s [%ix%i]\nread %s [%ix%i]\n", fx.c_str(), X.dim[0], X.dim[1], fh.c_str(), H.dim[0], H.dim[1],
           fw.c_str(), W.dim[0], W.dim[1]);
    nmf_result res;
    st = update_div_ex(W, H, X, &o, &res);
    if (st != NMF_OK) {
        fprintf(stderr, "nmf: update_div failed: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        return st;
    }
    if ((st = nmf_write_matrix(W, fwo.c_str())) || (st = nmf_write_matrix(H, fho.c_str()))) {
        fprintf(stderr, "nmf: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        return st;
    }
    printf("write %s [%ix%i]\nwrite %s [%ix%i]\n", fwo.c_str(), W.dim[0], W.dim[1], fho.c_str(), H.dim[0], H.dim[1]);
    printf("%d iterations, %.3f s total (%.3f s h2d, %.3f s d2h, %.3f s setup)\n", res.iterations, res.t[NMF_T_TOTAL], res.t[NMF_T_H2D],
           res.t[NMF_T_D2H], res.t[NMF_T_SETUP]);
    if (timers)
        printf("device time: H-step %.4f s, W-step %.4f s, sums %.4f s, apply %.4f s, checks %.4f s\n", res.t[NMF_T_H_STEP],
               res.t[NMF_T_W_STEP], res.t[NMF_T_SUMS], res.t[NMF_T_APPLY], res.t[NMF_T_CHECK]);
    nmf_destroy_matrix(&X); nmf_destroy_matrix(&W); nmf_destroy_matrix(&H);
    return 0;
}
