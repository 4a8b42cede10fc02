// nmf_cli.cpp -- command-line driver, the counterpart of the reference's main()
// (cuda/nmf.cu:30-51): read X, H, W (.bin), run update_div, write Wout, Hout.
// The reference hard-codes ../X.bin ../H.bin ../W.bin -> ../Wout.bin ../Hout.bin
// (cuda/nmf.cu:37-45) and compile-time MAX_ITER / CONVERGE_THRESH (cuda/nmf.cu:9-11);
// those stay the defaults, every one of them is a flag here.
#include "../../include/nmf_mi355x.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

static void usage(const char *argv0) {
    fprintf(stderr,
            "usage: %s [--X f] [--W f] [--H f] [--Wout f] [--Hout f] [--iters n] [--thresh x]\n"
            "          [--check n] [--verbose] [--timers] [--path auto|fused|unfused] [--device d]\n"
            "          [--devices n] (GPUs to shard the columns over; default: all when the problem is large enough)\n"
            "          [--restarts r] [--seed s] (paper section 3.2: r initialisations -- the W, H files, then r - 1 random\n"
            "           pairs from MT19937(s + i) -- against the one X; the pair with the lowest final KL divergence is written;\n"
            "           --devices n then deals the restarts to n GPUs, no communicator involved)\n"
            "       %s generate [--M m] [--N n] [--K k] [--seed s] [--X f] [--W f] [--H f]\n"
            "       %s compare A.bin B.bin [--tol t]\n"
            "defaults follow cuda/nmf.cu:9-11,37-45: ../X.bin ../W.bin ../H.bin -> ../Wout.bin ../Hout.bin,\n"
            "200 iterations, threshold 0, check every 25.  `generate` restates matrix_export.py (same bytes for the same\n"
            "seed and shape, default 4096 x 350, K = 128); `compare` is test_output.sh with a tolerance (rel-Frobenius of A\n"
            "against B, default 1e-4) instead of an md5.\n", argv0, argv0, argv0);
}

// matrix_export.py:4-7 restated: numpy.random.seed(s); rand(rows, cols).astype(float32) is MT19937 (init_genrand) with
// 53-bit doubles (a >> 5, b >> 6), drawn X -> W -> H from one stream; tobytes() of the C-ordered array is written
// behind the (rows, cols) header and later read as column-major (cuda/nmf.cu:194-204), which is what we reproduce.
static int write_random(std::mt19937 &g, const char *path, int rows, int cols) {
    matrix m;
    int st = nmf_create_matrix(&m, rows, cols, 0.0f);
    if (st != NMF_OK) return st;
    const size_t n = (size_t)rows * cols;
    for (size_t i = 0; i < n; ++i) {
        const unsigned a = (unsigned)g() >> 5, b = (unsigned)g() >> 6;
        m.mat[i] = (float)(((double)a * 67108864.0 + (double)b) / 9007199254740992.0);
    }
    st = nmf_write_matrix(m, path);
    if (st == NMF_OK) printf("wrote %s [%ix%i]\n", path, rows, cols);
    nmf_destroy_matrix(&m);
    return st;
}

static int cmd_generate(int argc, char **argv) {
    int M = 4096, N = 350, K = 128;
    unsigned seed = 0;
    std::string fx = "X.bin", fw = "W.bin", fh = "H.bin";
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { usage(argv[0]); exit(2); } return argv[++i]; };
        if (a == "--M") M = atoi(next());
        else if (a == "--N") N = atoi(next());
        else if (a == "--K") K = atoi(next());
        else if (a == "--seed") seed = (unsigned)strtoul(next(), nullptr, 10);
        else if (a == "--X") fx = next();
        else if (a == "--W") fw = next();
        else if (a == "--H") fh = next();
        else { usage(argv[0]); return 2; }
    }
    if (M <= 0 || N <= 0 || K <= 0) { fprintf(stderr, "nmf generate: M, N, K must be positive\n"); return 2; }
    std::mt19937 g(seed);
    int st;
    if ((st = write_random(g, fx.c_str(), M, N)) || (st = write_random(g, fw.c_str(), M, K)) || (st = write_random(g, fh.c_str(), K, N))) {
        fprintf(stderr, "nmf generate: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        return st;
    }
    return 0;
}

// test_output.sh compares md5s, which no two GEMM implementations can satisfy; this reports rel-Frobenius instead
static int cmd_compare(int argc, char **argv) {
    double tol = 1e-4;
    const char *fa = nullptr, *fb = nullptr;
    for (int i = 2; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--tol") { if (i + 1 >= argc) { usage(argv[0]); return 2; } tol = atof(argv[++i]); }
        else if (!fa) fa = argv[i];
        else if (!fb) fb = argv[i];
        else { usage(argv[0]); return 2; }
    }
    if (!fa || !fb) { usage(argv[0]); return 2; }
    matrix A, B;
    int st;
    if ((st = nmf_read_matrix(&A, fa)) || (st = nmf_read_matrix(&B, fb))) {
        fprintf(stderr, "nmf compare: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        return st;
    }
    if (A.dim[0] != B.dim[0] || A.dim[1] != B.dim[1]) {
        printf("%s is %ix%i, %s is %ix%i: different\n", fa, A.dim[0], A.dim[1], fb, B.dim[0], B.dim[1]);
        return 1;
    }
    const size_t n = (size_t)A.dim[0] * A.dim[1];
    double num = 0.0, den = 0.0, maxabs = 0.0;
    size_t nonfinite = 0;
    const bool same = memcmp(A.mat, B.mat, n * sizeof(float)) == 0;
    for (size_t i = 0; i < n; ++i) {
        const double a = A.mat[i], b = B.mat[i], d = a - b;
        if (!std::isfinite(a) || !std::isfinite(b)) { ++nonfinite; continue; }
        num += d * d; den += b * b;
        if (std::fabs(d) > maxabs) maxabs = std::fabs(d);
    }
    const double rel = den > 0.0 ? std::sqrt(num / den) : (num > 0.0 ? INFINITY : 0.0);
    printf("%s vs %s [%ix%i]: rel-Frobenius %.3e, max |a-b| %.3e, non-finite %zu, bytes %s\n", fa, fb, A.dim[0], A.dim[1], rel, maxabs,
           nonfinite, same ? "identical" : "differ");
    const bool ok = nonfinite == 0 && rel <= tol;
    printf(ok ? "Result matches the test data within %.1e\n" : "Result differs from the test data (tolerance %.1e)\n", tol);
    nmf_destroy_matrix(&A); nmf_destroy_matrix(&B);
    return ok ? 0 : 1;
}

int main(int argc, char **argv) {
    if (argc > 1 && !strcmp(argv[1], "generate")) return cmd_generate(argc, argv);
    if (argc > 1 && !strcmp(argv[1], "compare")) return cmd_compare(argc, argv);
    std::string fx = "../X.bin", fw = "../W.bin", fh = "../H.bin", fwo = "../Wout.bin", fho = "../Hout.bin";
    nmf_opts o;
    nmf_default_opts(&o);
    bool timers = false;
    int restarts = 1;
    unsigned seed = 0;
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
        else if (a == "--devices") o.n_devices = atoi(next());
        else if (a == "--emulate-shards") o.emulate_shards = atoi(next());
        else if (a == "--device") o.device = atoi(next());
        else if (a == "--restarts") restarts = atoi(next());
        else if (a == "--seed") seed = (unsigned)strtoul(next(), nullptr, 10);
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
    if (restarts > 1) {   // multi-restart NMF: the given pair first, then random ones; best final KL wins (update_div_restarts)
        std::vector<matrix> Ws((size_t)restarts), Hs((size_t)restarts);
        Ws[0] = W; Hs[0] = H;
        for (int r = 1; r < restarts; ++r) {
            if ((st = nmf_create_matrix(&Ws[(size_t)r], W.dim[0], W.dim[1], 0.f)) || (st = nmf_create_matrix(&Hs[(size_t)r], H.dim[0], H.dim[1], 0.f))) return st;
            std::mt19937 g(seed + (unsigned)r);
            auto fill = [&](matrix &m) {
                const size_t n = (size_t)m.dim[0] * m.dim[1];
                for (size_t i = 0; i < n; ++i) { const unsigned a = (unsigned)g() >> 5, b = (unsigned)g() >> 6; m.mat[i] = (float)(((double)a * 67108864.0 + (double)b) / 9007199254740992.0); }
            };
            fill(Ws[(size_t)r]); fill(Hs[(size_t)r]);
        }
        int best = -1;
        std::vector<double> kl((size_t)restarts);
        st = update_div_restarts(Ws.data(), Hs.data(), restarts, X, &o, &best, kl.data());
        if (st != NMF_OK) { fprintf(stderr, "nmf: update_div_restarts failed: %s (%s)\n", nmf_status_string(st), nmf_last_error()); return st; }
        for (int r = 0; r < restarts; ++r) printf("restart %d: kl-divergence %.9e%s\n", r, kl[(size_t)r], r == best ? "  <- best" : "");
        if ((st = nmf_write_matrix(Ws[(size_t)best], fwo.c_str())) || (st = nmf_write_matrix(Hs[(size_t)best], fho.c_str()))) {
            fprintf(stderr, "nmf: %s (%s)\n", nmf_status_string(st), nmf_last_error());
            return st;
        }
        printf("write %s [%ix%i]\nwrite %s [%ix%i]\n", fwo.c_str(), W.dim[0], W.dim[1], fho.c_str(), H.dim[0], H.dim[1]);
        for (int r = 0; r < restarts; ++r) { nmf_destroy_matrix(&Ws[(size_t)r]); nmf_destroy_matrix(&Hs[(size_t)r]); }
        nmf_destroy_matrix(&X);
        return 0;
    }
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
