/*
 * nmf_oracle.h -- CPU oracle for the update_div hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is a plain-C restatement of the reference's KL-divergence multiplicative-update
 * NMF loop.  It is the parity checker for the HIP path and the timed CPU baseline
 * ("port") of bench.py.  Nothing in the shipped product (nmf-gpu_amd/) may include,
 * link, import or call it; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Parity status: PINNED for `refcompat` mode by the reference's own golden outputs
 * Wtest.bin / Htest.bin (tests/golden/, see tests/test_oracle_golden.py); `spec` mode
 * (the intended math, which the GPU path implements) shares every arithmetic routine
 * with `refcompat` and differs only in (i) using all 128 per-thread partials of the
 * column sum and (ii) applying the row_divide the reference silently drops.
 *
 * All matrices are column-major, leading dimension = rows, fp32, exactly as the
 * reference's .bin files (cuda/nmf.cu:188-259, README.md:31-36).
 */
#ifndef NMF_ORACLE_H
#define NMF_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* EPS exactly as cuda/matrix.cu:10 */
#define ORACLE_EPS ((float)(2.2204E-16))

enum { ORACLE_MODE_SPEC = 0, ORACLE_MODE_REFCOMPAT = 1 };

/* numpy.random.seed(seed); numpy.random.rand(n).astype(float32) restated
 * (matrix_export.py:4-7).  `state` is an opaque 625-word MT19937 state so that the
 * X, W, H draws can be chained from one stream like the script does. */
typedef struct { uint32_t mt[624]; int idx; } oracle_rng;
void   oracle_rng_seed(oracle_rng *g, uint32_t seed);
void   oracle_rng_fill_f32(oracle_rng *g, float *out, size_t n);

/* clamp: a[i] < EPS -> EPS   (cuda/matrix.cu:182-188, input clamp cuda/nmf.cu:211) */
void   oracle_set_epsilon(float *a, size_t n);

/* C = A*B, C = A'*B, C = A*B'   (cuda/matrix.cu:97-125), column-major, fp32 */
void   oracle_sgemm_nn(int m, int n, int k, const float *A, const float *B, float *C);
void   oracle_sgemm_tn(int m, int n, int k, const float *A, const float *B, float *C);
void   oracle_sgemm_nt(int m, int n, int k, const float *A, const float *B, float *C);

/* column sums of an (rows x cols) matrix -> out[cols]   (cuda/matrix.cu:642-687)
 * row sums                                 -> out[rows]   (cuda/matrix.cu:689-735) */
void   oracle_sum_cols(const float *A, int rows, int cols, float *out);
void   oracle_sum_rows(const float *A, int rows, int cols, float *out);
/* the reference's miscompiled reduce2d<128> as pinned by Wtest/Htest */
void   oracle_sum_cols_refcompat(const float *A, int rows, int cols, float *out);

/* KL divergence  sum x*(log x - log y) - x + y   (cuda/matrix.cu:592), fp64 accumulate */
double oracle_kl_div(const float *X, const float *Y, size_t n);
/* rel-L1 error  sum|x-y| / sum|x|   (cuda/matrix.cu:517-518) */
double oracle_rel_l1(const float *X, const float *Y, size_t n);

/* One H half-step / W half-step (cuda/nmf.cu:118-146 / 148-176).  Z is M*N scratch. */
void   oracle_update_h(float *W, float *H, const float *X, int M, int N, int K,
                       float *Z, float *WtZ, float *sumW, int mode);
void   oracle_update_w(float *W, float *H, const float *X, int M, int N, int K,
                       float *Z, float *ZHt, float *sumH, int mode);

/* The loop (cuda/nmf.cu:76-116 + README.md:40-54 convergence contract).
 *  - W (M x K), H (K x N) in/out; X (M x N) read-only; inputs are clamped to EPS on
 *    private copies of X and in place on W,H (cuda/nmf.cu:211).
 *  - iter_check > 0: every iter_check iterations compute KL(X || W*H); stop when
 *    (prev-cur)/prev < thresh (thresh == 0 -> never stop early).
 *  - kl_trace (may be NULL): receives KL at iteration 0 (before any update) and at each
 *    check, up to kl_cap entries; *n_kl gets the count.
 * Returns the number of iterations actually executed. */
int    oracle_update_div(float *W, float *H, const float *X, int M, int N, int K,
                         float thresh, int max_iter, int iter_check, int mode,
                         double *kl_trace, int kl_cap, int *n_kl);

/* .bin file format: uint32 rows, uint32 cols, float32[rows*cols] column-major
 * (cuda/nmf.cu:194-204, 239-249).  read returns malloc'd data (caller frees). */
int    oracle_read_bin(const char *path, uint32_t *rows, uint32_t *cols, float **data);
int    oracle_write_bin(const char *path, uint32_t rows, uint32_t cols, const float *data);

/* The timed CPU baseline (nmf_oracle_fast.c): `iters` spec-mode iterations with register-tiled SGEMM micro-kernels.
 * Same math, different summation order than oracle_update_div -- checked against it by tolerance, never used as
 * the parity checker.  Returns 0, -1 if scratch cannot be allocated. */
int    oracle_fast_update_div(float *W, float *H, const float *X, int M, int N, int K, int iters);

int    oracle_num_threads(void);
void   oracle_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
