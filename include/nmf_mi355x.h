/*
 * nmf_mi355x.h -- C ABI of libnmf_mi355x.so: the MI355X (gfx950) implementation of the
 * reference's KL-divergence multiplicative-update NMF hot path (update_div, X ~ W*H).
 *
 * Every entry point cites the reference interface it replaces (paths relative to the
 * reference repository root).  Plain pointers and sizes only; no torch / C++ types.
 * All matrices are fp32, column-major, leading dimension = rows (README.md:31-36).
 */
#ifndef NMF_MI355X_H
#define NMF_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* the shared library is built with -fvisibility=hidden: only what this header declares is exported */
#pragma GCC visibility push(default)

/* ---------------------------------------------------------------------------------------
 * The reference's matrix struct (README.md:31-36: "column-major float array ... proper
 * assignment of the dim values"; upstream layout {mat, mat_d, dim[2]}).  `mat` is the
 * host buffer (dim[0] x dim[1], ld = dim[0], unpadded); `mat_d` an optional device buffer
 * of the same unpadded layout (NULL when absent).  Replaces class Matrix,
 * cuda/matrix.cuh:18-39.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    float *mat;
    float *mat_d;
    int    dim[2];
} matrix;

/* EPS exactly as cuda/matrix.cu:10 */
#define NMF_EPS ((float)(2.2204E-16))
/* cuda/nmf.cu:9 */
#define NMF_ITER_CHECK_DEFAULT 25
/* nmf_opts.use_graph: let the library decide between hipGraph replay and eager launches */
#define NMF_GRAPH_AUTO 2

/* status codes (the reference has none: it prints and exit()s, error-check.hpp:12-17) */
enum {
    NMF_OK = 0,
    NMF_ERR_ARG = 1,        /* NULL pointer / non-positive dimension */
    NMF_ERR_SHAPE = 2,      /* "dimensions do not agree" (cuda/matrix.cu:130-134 etc.) */
    NMF_ERR_HIP = 3,        /* a HIP runtime call or kernel launch failed */
    NMF_ERR_IO = 4,         /* fopen / fread / fwrite failure (cuda/nmf.cu:196-205) */
    NMF_ERR_NOMEM = 5,
    NMF_ERR_COMM = 6,       /* RCCL failure, or a collective that ran into its deadline (NMF_COMM_TIMEOUT_S, default 30 s) */
    NMF_ERR_UNSUPPORTED = 7
};

/* which device path update_div runs */
enum {
    NMF_PATH_AUTO = 0,      /* fused when K <= 1024, else unfused */
    NMF_PATH_FUSED = 1,     /* two fused MFMA kernels per iteration, Z never reaches HBM */
    NMF_PATH_UNFUSED = 2    /* op-for-op twin of cuda/nmf.cu:118-176 (16 kernels/iteration) */
};

/* meaning of the README's `double t[10]` (README.md:53; slot meaning is undefined in the
 * reference tree, defined here).  All values in seconds, accumulated over the call. */
enum {
    NMF_T_TOTAL = 0,        /* whole call, wall clock */
    NMF_T_H2D = 1,          /* upload + clamp of W, H, X */
    NMF_T_H_STEP = 2,       /* H half-steps (device time) */
    NMF_T_W_STEP = 3,       /* W half-steps (device time) */
    NMF_T_SUMS = 4,         /* column/row-sum normalisers */
    NMF_T_APPLY = 5,        /* split-partial reduce + multiplicative apply */
    NMF_T_CHECK = 6,        /* KL-divergence convergence checks */
    NMF_T_ALLREDUCE = 7,    /* RCCL all-reduce (sharded runs) */
    NMF_T_D2H = 8,          /* download of W, H */
    NMF_T_SETUP = 9         /* allocation + hipGraph capture/instantiate */
};

/* ---------------------------------------------------------------------------------------
 * update_div -- the documented drop-in surface (README.md:40-54); body = run_async,
 * cuda/nmf.cu:76-116.  W (M x K) and H (K x N) are in/out through W.mat / H.mat (host);
 * X (M x N) is read-only.  Inputs are clamped to >= EPS on the device copies only
 * (cuda/nmf.cu:210-211).  Every NMF_ITER_CHECK_DEFAULT iterations the KL divergence
 * (cuda/matrix.cu:592) is evaluated; the loop stops when (prev-cur)/prev <
 * CONVERGE_THRESH; CONVERGE_THRESH == 0 runs exactly max_iter iterations and skips the
 * checks unless verbose (cuda/nmf.cu:9-11).  t may be NULL.  Errors: message on stderr +
 * exit(code), as error-check.hpp:12-17 / cuda/matrix.cu:130-134.
 * ------------------------------------------------------------------------------------- */
void update_div(matrix W, matrix H, matrix X, float CONVERGE_THRESH, int max_iter, double t[10],
                int verbose);

typedef struct {
    float converge_thresh;  /* README.md:51 */
    int   max_iter;         /* cuda/nmf.cu:10 */
    int   iter_check;       /* cuda/nmf.cu:9; <= 0 -> NMF_ITER_CHECK_DEFAULT */
    int   verbose;          /* README.md:54 */
    int   path;             /* NMF_PATH_* */
    int   use_graph;        /* NMF_GRAPH_AUTO (default): the resident solver replays captured hipGraphs; a one-shot update_div /
                             * update_div_ex call does too unless the whole run is shorter than capturing is worth (8*M*N*K*max_iter
                             * < 2e12 flop), where it launches eagerly; 1: replay captured hipGraphs of 32 / 8 / 1 iterations
                             * (cuda/nmf.cu:100-115); 0: eager launches with a hipEvent pair around every piece (fills t[2..7]);
                             * -1: eager launches, untimed */
    int   device;           /* HIP device ordinal, -1 = current */
    void *stream;           /* hipStream_t to run on, NULL = library-owned stream */
    void *comm;             /* nmf_comm* for an N-sharded run (see nmf_comm_*), NULL = single GPU */
    int   nsplit_h;         /* 0 = auto; split count of the reduction dim in the H / W step */
    int   nsplit_w;
    int   fast_divide;      /* 0 (default): X ./ WH correctly rounded (IEEE division, cuda/matrix.cu:149).  The fused kernel
                             *    omits the range-scaling steps of the division sequence for waves whose operands are all
                             *    in [EPS, 2^60] (X checked at upload, W*H per 32-row chunk) -- there they are the identity,
                             *    so the quotient is bit-identical -- and runs the full sequence everywhere else;
                             * -1: always the full sequence (for A/B tests of the above);
                             * 1: accepted and ignored since round 5 (it selected the same six-instruction quotient without the range
                             *    guard -- bit-identical inside the guard range by exhaustive enumeration, DESIGN.md 4.1c -- and measured
                             *    0.1 .. 1.9 % slower than the guarded default: profiles/r05_ab_fast_divide.log) */
    int   restart_lanes;    /* update_div_restarts only.  0 = automatic: shapes the split kernel takes (K <= 256 and small enough, see split_kernel)
                             * and shapes of the 64-column kernel whose lone launch leaves CUs idle (K <= 512, N < 32768) run ALL restarts in
                             * every launch (the restart index is a grid dimension); other shapes iterate two initialisations side by side on
                             * their own streams unless one launch already fills the chip.
                             * n > 0: n stream lanes (1 = one restart after the other), never the batched grid */
    int   split_kernel;     /* which fused kernel family: 0 = automatic; 1 = the split kernel (four waves per 16 owned columns,
                             * normalisers summed in-stream, two to four launches per iteration: problems that do not fill
                             * the chip, K <= 256); -1 = never (the 64-column kernel of the large configurations) */
    int   n_devices;        /* update_div / update_div_ex with host matrices: how many GPUs of this node share the work (columns of
                             * X and H are sharded, W is replicated, one RCCL all-reduce of [Z*H' ; rowsum(H)] per iteration inside
                             * each device's hipGraph; one host thread per device, all inside the call).  SEVERAL GPUS ARE OPT-IN: the
                             * N > 1 path is tested with emulated ranks and one-rank RCCL communicators only -- no multi-GPU node has run
                             * it yet -- so 0 (default) = ONE GPU unless the environment asks (NMF_DEVICES=<n>|all forces a count,
                             * NMF_DEVICES=auto takes every visible device when nmf_worth_sharding says the problem amortises the
                             * all-reduce) and `device` does not pin one; 1 = one; n > 1 = exactly n.  Every wait of a sharded run has a
                             * deadline (NMF_COMM_TIMEOUT_S, default 30 s for the first collective): a rank that never arrives gets the
                             * group aborted and the call NMF_ERR_COMM; when the request came from the environment the call then runs on
                             * one GPU instead (W.mat / H.mat are written only by a run that succeeded on every rank).
                             * update_div_restarts: the number of workers the restarts are dealt to */
    const int *devices;     /* optional list of n_devices HIP ordinals; NULL = device, device + 1, ... (device < 0: from 0).  With
                             * n_devices = 1 an explicit list still takes the multi-device driver (one thread, one RCCL rank) */
    int   emulate_shards;   /* G > 1: run the multi-device driver with G ranks on ONE device, the all-reduce replaced by a
                             * device-side sum in rank order (for one-GPU test boxes; at most 8) */
} nmf_opts;

#define NMF_MAX_KL 64
typedef struct {
    int    iterations;          /* iterations actually executed */
    int    n_kl;                /* KL values recorded (iteration 0 first when checks are on) */
    double kl[NMF_MAX_KL];
    double rel_l1;              /* sum|X-WH| / sum|X| at the last check (cuda/matrix.cu:517-518) */
    int    path_used;
    double t[10];
    int    n_shards;            /* ranks the columns were sharded over (1 = single GPU) */
    int    w_replicas_identical;/* sharded runs: 1 if every rank ended with bit-identical W (it must), else 0; single GPU: 1 */
} nmf_result;

void nmf_default_opts(nmf_opts *o);
/* same contract as update_div but returns a status instead of exiting */
int  update_div_ex(matrix W, matrix H, matrix X, const nmf_opts *opts, nmf_result *res);
const char *nmf_status_string(int status);
/* message of the last failing call on this thread (file:line + HIP error string) */
const char *nmf_last_error(void);

/* ---------------------------------------------------------------------------------------
 * matrix helpers: read_matrix / write_matrix (cuda/nmf.cu:188-259) and the Matrix
 * constructors / destructor (cuda/matrix.cu:42-86).  File format: little-endian
 * uint32 rows, uint32 cols, float32[rows*cols] column-major (cuda/nmf.cu:194-204).
 * ------------------------------------------------------------------------------------- */
int  nmf_create_matrix(matrix *A, int rows, int cols, float value);   /* host alloc + fill */
void nmf_destroy_matrix(matrix *A);                                   /* frees mat and mat_d */
int  nmf_read_matrix(matrix *A, const char *file);                    /* cuda/nmf.cu:188-218 (no clamp: host copy stays raw) */
int  nmf_write_matrix(matrix A, const char *file);                    /* cuda/nmf.cu:220-259 */
int  nmf_matrix_alloc_device(matrix *A, int rows, int cols);   /* device buffer only, contents undefined (Matrix(rows, cols), cuda/matrix.cu:42-51) */
int  nmf_matrix_to_device(matrix *A);      /* allocates mat_d if NULL, H2D (cuda/matrix.cu:53-67) */
int  nmf_matrix_from_device(matrix *A);    /* D2H into mat (cuda/nmf.cu:228-232) */
int  nmf_matrix_free_device(matrix *A);

/* ---------------------------------------------------------------------------------------
 * Device operators, one per reference operator (cuda/matrix.cuh:41-52).  They work on
 * the mat_d buffers of their arguments (unpadded, ld = dim[0]) on `stream` (hipStream_t,
 * NULL = default stream), check shapes like the reference and return NMF_ERR_SHAPE where
 * the reference exit(1)s.  Used by the NMF_PATH_UNFUSED loop and by the parity tests.
 * ------------------------------------------------------------------------------------- */
int nmf_matrix_multiply(matrix a, matrix b, matrix c, void *stream);      /* c = a*b    cuda/matrix.cu:97-105  */
int nmf_matrix_multiply_AtB(matrix a, matrix b, matrix c, void *stream);  /* c = a'*b   cuda/matrix.cu:107-115 */
int nmf_matrix_multiply_ABt(matrix a, matrix b, matrix c, void *stream);  /* c = a*b'   cuda/matrix.cu:117-125 */
int nmf_element_multiply(matrix a, matrix b, matrix c, void *stream);     /* c = a.*b   cuda/matrix.cu:154-180 */
int nmf_element_divide(matrix a, matrix b, matrix c, void *stream);       /* c = a./b   cuda/matrix.cu:127-152 */
int nmf_row_divide(matrix a, matrix b, matrix c, void *stream);           /* c[i,j] = a[i,j]/b[j]  cuda/matrix.cu:203-224 */
int nmf_col_divide(matrix a, matrix b, matrix c, void *stream);           /* c[i,j] = a[i,j]/b[i]  cuda/matrix.cu:226-250 */
int nmf_set_epsilon(matrix a, void *stream);                              /* a<EPS -> EPS  cuda/matrix.cu:182-201 */
int nmf_sum_cols(matrix a, matrix out, void *stream);                     /* out(1 x cols)  cuda/matrix.cu:261-377,642-687 */
int nmf_sum_rows(matrix a, matrix out, void *stream);                     /* out(rows x 1)  cuda/matrix.cu:379-503,689-735 */
/* KL divergence sum x(log x - log y) - x + y (reduce1d_div, cuda/matrix.cu:578-640) and
 * sum|x-y|, sum|x| (reduce1d_diff, cuda/matrix.cu:505-576); synchronises `stream`. */
int nmf_kl_divergence(matrix x, matrix y, double *kl, void *stream);
int nmf_diff_norm(matrix x, matrix y, double *sum_abs_diff, double *sum_abs_x, void *stream);

/* ---------------------------------------------------------------------------------------
 * Resident solver: what update_div does between its H2D and D2H, exposed so that a
 * caller (bench.py, the CLI, a multi-GPU launcher) can keep W, H, X in HBM across calls.
 * Replaces run_async + its temporaries (cuda/nmf.cu:76-116).
 * ------------------------------------------------------------------------------------- */
typedef struct nmf_solver nmf_solver;

/* M, N, K are the LOCAL problem dims of this process (N = this rank's column count when
 * opts->comm is set). */
int  nmf_solver_create(nmf_solver **s, int M, int N, int K, const nmf_opts *opts);
void nmf_solver_destroy(nmf_solver *s);
/* host -> device (clamped to EPS like read_matrix, cuda/nmf.cu:211); any pointer may be NULL to skip */
int  nmf_solver_upload(nmf_solver *s, const float *W, const float *H, const float *X);
/* same from unpadded device buffers (ld = rows) */
int  nmf_solver_upload_device(nmf_solver *s, const float *W_d, const float *H_d, const float *X_d);
int  nmf_solver_download(nmf_solver *s, float *W, float *H);
/* enqueue `iters` iterations (H half-step then W half-step each, cuda/nmf.cu:108-109) on the
 * solver's stream; does not synchronise. */
int  nmf_solver_iterate(nmf_solver *s, int iters);
/* capture and instantiate the hipGraphs that nmf_solver_iterate(s, iters) would replay, without running them (keeps the
 * one-off capture cost out of a caller's timed region) */
int  nmf_solver_prepare(nmf_solver *s, int iters);
/* `iters` iterations launched eagerly with a hipEvent pair around every piece (the README's t[10], README.md:53):
 * adds the device seconds of each piece to t[NMF_T_H_STEP .. NMF_T_ALLREDUCE]; synchronises. */
int  nmf_solver_iterate_timed(nmf_solver *s, int iters, double t[10]);
/* the two half-steps separately (cuda/nmf.cu:118-146 / 148-176) */
int  nmf_solver_update_h(nmf_solver *s);
int  nmf_solver_update_w(nmf_solver *s);
/* KL(X || W*H) and rel-L1 of the current state; synchronises.  In a sharded run the values
 * are summed over ranks. */
int  nmf_solver_check(nmf_solver *s, double *kl, double *rel_l1);
/* the three raw sums behind the check: {KL, sum|X-WH|, sum|X|}; local to this rank unless opts->comm is set */
int  nmf_solver_check_sums(nmf_solver *s, double sums[3]);
/* full loop with convergence logic; fills res (may be NULL) */
int  nmf_solver_run(nmf_solver *s, float thresh, int max_iter, int iter_check, int verbose, nmf_result *res);
int  nmf_solver_sync(nmf_solver *s);
/* time one piece in isolation: which = NMF_T_H_STEP / NMF_T_W_STEP / NMF_T_SUMS / NMF_T_APPLY /
 * NMF_T_CHECK; runs it `reps` times between two hipEvents on the solver's stream and returns the
 * average milliseconds per launch of the dominant kernel of that piece.  Any other `which` is NMF_ERR_ARG in the
 * shipped library; a diagnostic build (make DIAG=1 in nmf-gpu_amd/csrc) adds the measurement probes of tools/probe.py
 * and tools/divide_exhaustive.py there. */
int  nmf_solver_time_piece(nmf_solver *s, int which, int reps, double *ms_per_launch);
/* sharded runs where the caller performs the all-reduce itself (e.g. torch.distributed):
 * w_partial leaves sum_g-local [Z*H' (M*K floats) ; rowsum(H) (K floats)] in a device buffer
 * (pointer/count returned by nmf_solver_partial_buffer); after reducing it in place across ranks
 * call w_apply. */
int  nmf_solver_w_partial(nmf_solver *s);
int  nmf_solver_w_apply(nmf_solver *s);
int  nmf_solver_partial_buffer(nmf_solver *s, float **dev_ptr, size_t *count);
/* make the solver use a caller-owned device buffer of `count` floats (>= the count reported above) as
 * its partial buffer, e.g. the storage of a torch tensor that torch.distributed will all-reduce */
int  nmf_solver_set_partial_buffer(nmf_solver *s, float *dev_ptr, size_t count);
/* B independent (W, H) pairs against one X in every launch (multi-restart NMF, paper section 3.2; the restart index is a
 * grid dimension of the split kernel, K <= 256).  Pair b is addressed by the *_pair calls; upload / download / check
 * without a pair index act on pair 0.  `flags` (batch ints, host) freezes pairs whose flag is 0: they are skipped by
 * every later iterate (a converged restart stops exactly where a sequential update_div would). */
int  nmf_solver_create_batched(nmf_solver **s, int M, int N, int K, int batch, const nmf_opts *opts);
int  nmf_solver_batch(const nmf_solver *s);
int  nmf_solver_upload_pair(nmf_solver *s, int b, const float *W, const float *H);
int  nmf_solver_download_pair(nmf_solver *s, int b, float *W, float *H);
int  nmf_solver_check_pair(nmf_solver *s, int b, double *kl, double *rel_l1);
int  nmf_solver_set_active(nmf_solver *s, const int *flags);
/* KL and rel-L1 of every pair (arrays of `batch` doubles, either may be NULL) behind one synchronisation */
int  nmf_solver_check_all(nmf_solver *s, double *kl, double *rel_l1);
/* 1 if this solver runs the split kernel (see nmf_opts.split_kernel) */
int  nmf_solver_uses_split_kernel(const nmf_solver *s);
int  nmf_solver_path(const nmf_solver *s);
/* one line: kernel family, padded shape, split counts (for logs and bench records) */
int  nmf_solver_describe(const nmf_solver *s, char *buf, int buflen);
void *nmf_solver_stream(nmf_solver *s);

/* ---------------------------------------------------------------------------------------
 * Multi-restart NMF (paper section 3.2: "multiple random initializations and choose the best"): X is
 * uploaded once, each (W[i], H[i]) pair runs the same update_div loop, the pair with the lowest final KL
 * divergence wins.  All pairs are updated in place; *best receives the winner's index, kl[i] (may be NULL)
 * each pair's final KL.
 * Several GPUs ("replicas only", below the size where sharding ONE problem pays): restart i runs on worker i % G, one host
 * thread + one batched solver + one copy of X per worker, no communicator and no collective.  opts->n_devices = G > 1
 * (optionally with opts->devices, which may name a device more than once) forces G workers; 0 = one device unless the
 * environment asks (NMF_DEVICES=<n>|all; NMF_DEVICES=auto: every visible device when the call holds at least ~0.5 s of
 * single-GPU work, 5e13 flop -- bringing up a device costs a few hundred milliseconds) and opts->device does not pin one;
 * 1 = one device.  Needs X.mat (host).  Every restart
 * gets the same kernels and split counts wherever it runs: its result is bit-identical to the one-device call's.
 * ------------------------------------------------------------------------------------- */
int  update_div_restarts(const matrix *W, const matrix *H, int n_restarts, matrix X, const nmf_opts *opts,
                         int *best, double *kl);

/* ---------------------------------------------------------------------------------------
 * RCCL communicator for N-sharded runs (new: the reference is single-GPU).  One process
 * per GPU; rank 0 creates the id, the launcher broadcasts its 128 bytes (e.g. with
 * torch.distributed), every rank calls nmf_comm_init_rank.
 * ------------------------------------------------------------------------------------- */
typedef struct nmf_comm nmf_comm;
#define NMF_COMM_ID_BYTES 128
int  nmf_comm_get_unique_id(unsigned char id[NMF_COMM_ID_BYTES]);
int  nmf_comm_init_rank(nmf_comm **c, const unsigned char id[NMF_COMM_ID_BYTES], int rank, int nranks);
void nmf_comm_destroy(nmf_comm *c);
/* Collective (every rank of the communicator calls it): one 8-float all-reduce waited for with a deadline (timeout_s <= 0:
 * NMF_COMM_TIMEOUT_S, default 30 s) and checked.  NMF_OK, or NMF_ERR_COMM after the communicator has been aborted: a launcher that
 * can still fall back to another all-reduce (bench.py: torch.distributed) probes before it commits to this one. */
int  nmf_comm_probe(nmf_comm *c, double timeout_s);
/* one line: version and path of the RCCL library actually loaded, and the rccl.h version this library was compiled against
 * (a different major version is refused at load) */
int  nmf_comm_library_info(char *buf, int buflen);
/* Diagnostics.  A wait on a sharded run's stream makes no HIP call while it waits (it watches a pinned word the device writes behind
 * the awaited work) and closes with one hipStreamSynchronize whose status is final -- except hipErrorStreamCaptureUnsupported, which
 * reports a stream capture elsewhere in the process rather than this stream and is asked again up to three times.  This counts those
 * (process-wide); the library serialises its own captures, so the expected value is 0. */
long nmf_comm_capture_refusals(void);

/* 1 if sharding an M x N x K problem over n_devices GPUs amortises the per-iteration all-reduce (what NMF_DEVICES=auto uses) */
int  nmf_worth_sharding(int M, int N, int K, int n_devices);

/* diagnostics: with recording on, every launcher of a fused half-step / check kernel notes the demangled name of the
 * instantiation it launches (thread-local); nmf_debug_last_kernel returns the most recent one ("" before the first).
 * Used by the tests to hit every instantiation by name; off by default (one relaxed load per launch). */
int  nmf_debug_record_kernels(int on);     /* returns the previous setting */
const char *nmf_debug_last_kernel(void);

/* what nmf_solver_create_batched(M, N, K, batch, opts) would run -- kernel family, padded shape, split counts, as
 * nmf_solver_describe prints them -- without creating anything and without touching a device (works on a machine with no GPU) */
int  nmf_plan_describe(int M, int N, int K, int batch, const nmf_opts *opts, char *buf, int buflen);

/* device queries used by bench/tests */
int  nmf_device_count(void);
int  nmf_device_name(int device, char *buf, int buflen);
const char *nmf_version(void);

#pragma GCC visibility pop

#ifdef __cplusplus
}
#endif
#endif /* NMF_MI355X_H */
