/*
 * nmf_oracle.c -- CPU oracle for the update_div hot path.  TEST INFRASTRUCTURE ONLY:
 * see nmf_oracle.h for who may use it.  Plain C11 + OpenMP, no BLAS dependency.
 *
 * Every routine cites the reference lines it restates (paths relative to the
 * reference repository root).  Nothing here is copied from the reference: the
 * reference is CUDA + cuBLAS; this is a from-scratch host implementation of the same
 * arithmetic, in the same order of operations per element, with its own blocked SGEMMs.
 */
#include "nmf_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* The GPU box exposes every host CPU but grants a cgroup quota of a few: the Python wrapper
 * passes the quota here so OpenMP does not oversubscribe it. */
void oracle_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ RNG ------
 * matrix_export.py:4-7 uses numpy's legacy global RandomState: MT19937 seeded by
 * init_genrand(seed); rand() = 53-bit double from two 32-bit outputs; then
 * .astype(float32) (round-to-nearest-even double->float, which is the C cast). */
void oracle_rng_seed(oracle_rng *g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; i++)
        g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t rng_next_u32(oracle_rng *g) {
    if (g->idx >= 624) {
        uint32_t *mt = g->mt;
        for (int i = 0; i < 624; i++) {
            uint32_t y = (mt[i] & 0x80000000u) | (mt[(i + 1) % 624] & 0x7fffffffu);
            uint32_t v = mt[(i + 397) % 624] ^ (y >> 1);
            if (y & 1u) v ^= 0x9908b0dfu;
            mt[i] = v;
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void oracle_rng_fill_f32(oracle_rng *g, float *out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        uint32_t a = rng_next_u32(g) >> 5, b = rng_next_u32(g) >> 6;
        double d = ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
        out[i] = (float)d;
    }
}

/* ------------------------------------------------------------ elementwise ---- */
/* cuda/matrix.cu:182-188: `if (a[i] < EPS) a[i] = EPS` -- a clamp, not an add;
 * NaN compares false and stays NaN. */
void oracle_set_epsilon(float *a, size_t n) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++)
        if (a[i] < ORACLE_EPS) a[i] = ORACLE_EPS;
}

/* ------------------------------------------------------------------ SGEMMs ---
 * Column-major, alpha=1, beta=0 like the three cublasSgemm call sites
 * (cuda/matrix.cu:101-104, 111-114, 121-124).  fp32 multiply-add throughout. */

/* C(m x n) = A(m x k) * B(k x n) */
void oracle_sgemm_nn(int m, int n, int k, const float *restrict A, const float *restrict B,
                     float *restrict C) {
    const int IB = 1024;
#pragma omp parallel for schedule(dynamic, 1)
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int nj = (n - j0 < 4) ? (n - j0) : 4;
        for (int i0 = 0; i0 < m; i0 += IB) {
            const int ni = (m - i0 < IB) ? (m - i0) : IB;
            float *c0 = C + (size_t)(j0 + 0) * m + i0;
            float *c1 = C + (size_t)(j0 + (nj > 1 ? 1 : 0)) * m + i0;
            float *c2 = C + (size_t)(j0 + (nj > 2 ? 2 : 0)) * m + i0;
            float *c3 = C + (size_t)(j0 + (nj > 3 ? 3 : 0)) * m + i0;
            float t0[1024], t1[1024], t2[1024], t3[1024];
            memset(t0, 0, sizeof(float) * ni); memset(t1, 0, sizeof(float) * ni);
            memset(t2, 0, sizeof(float) * ni); memset(t3, 0, sizeof(float) * ni);
            for (int l = 0; l < k; l++) {
                const float *a = A + (size_t)l * m + i0;
                const float b0 = B[(size_t)(j0 + 0) * k + l];
                const float b1 = B[(size_t)(j0 + (nj > 1 ? 1 : 0)) * k + l];
                const float b2 = B[(size_t)(j0 + (nj > 2 ? 2 : 0)) * k + l];
                const float b3 = B[(size_t)(j0 + (nj > 3 ? 3 : 0)) * k + l];
                for (int i = 0; i < ni; i++) {
                    const float av = a[i];
                    t0[i] += av * b0; t1[i] += av * b1; t2[i] += av * b2; t3[i] += av * b3;
                }
            }
            /* write highest alias last so that nj<4 duplicates resolve to column j0 */
            if (nj > 3) memcpy(c3, t3, sizeof(float) * ni);
            if (nj > 2) memcpy(c2, t2, sizeof(float) * ni);
            if (nj > 1) memcpy(c1, t1, sizeof(float) * ni);
            memcpy(c0, t0, sizeof(float) * ni);
        }
    }
}

/* C(m x n) = A'(m x k) * B(k x n), A stored (k x m): C[i,j] = dot(A[:,i], B[:,j]) */
void oracle_sgemm_tn(int m, int n, int k, const float *restrict A, const float *restrict B,
                     float *restrict C) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int nj = (n - j0 < 4) ? (n - j0) : 4;
        for (int i0 = 0; i0 < m; i0 += 4) {
            const int ni = (m - i0 < 4) ? (m - i0) : 4;
            float acc[4][4] = {{0}};
            const float *a[4], *b[4];
            for (int q = 0; q < 4; q++) {
                a[q] = A + (size_t)(i0 + (q < ni ? q : 0)) * k;
                b[q] = B + (size_t)(j0 + (q < nj ? q : 0)) * k;
            }
            /* 8-lane partial sums so the compiler can vectorise the reduction */
            float p[4][4][8];
            memset(p, 0, sizeof p);
            int l = 0;
            for (; l + 8 <= k; l += 8)
                for (int ii = 0; ii < 4; ii++)
                    for (int jj = 0; jj < 4; jj++)
                        for (int v = 0; v < 8; v++)
                            p[ii][jj][v] += a[ii][l + v] * b[jj][l + v];
            for (int ii = 0; ii < 4; ii++)
                for (int jj = 0; jj < 4; jj++) {
                    float s = ((p[ii][jj][0] + p[ii][jj][4]) + (p[ii][jj][2] + p[ii][jj][6])) +
                              ((p[ii][jj][1] + p[ii][jj][5]) + (p[ii][jj][3] + p[ii][jj][7]));
                    for (int r = l; r < k; r++) s += a[ii][r] * b[jj][r];
                    acc[ii][jj] = s;
                }
            for (int jj = 0; jj < nj; jj++)
                for (int ii = 0; ii < ni; ii++)
                    C[(size_t)(j0 + jj) * m + i0 + ii] = acc[ii][jj];
        }
    }
}

/* C(m x n) = A(m x k) * B'(k x n), B stored (n x k): C[i,j] = sum_l A[i,l] * B[j,l] */
void oracle_sgemm_nt(int m, int n, int k, const float *restrict A, const float *restrict B,
                     float *restrict C) {
    const int IB = 256;
#pragma omp parallel for schedule(dynamic, 1)
    for (int i0 = 0; i0 < m; i0 += IB) {
        const int ni = (m - i0 < IB) ? (m - i0) : IB;
        for (int j = 0; j < n; j++) memset(C + (size_t)j * m + i0, 0, sizeof(float) * ni);
        int l0 = 0;
        for (; l0 + 4 <= k; l0 += 4) {
            const float *a0 = A + (size_t)(l0 + 0) * m + i0, *a1 = A + (size_t)(l0 + 1) * m + i0;
            const float *a2 = A + (size_t)(l0 + 2) * m + i0, *a3 = A + (size_t)(l0 + 3) * m + i0;
            for (int j = 0; j < n; j++) {
                const float b0 = B[(size_t)(l0 + 0) * n + j], b1 = B[(size_t)(l0 + 1) * n + j];
                const float b2 = B[(size_t)(l0 + 2) * n + j], b3 = B[(size_t)(l0 + 3) * n + j];
                float *c = C + (size_t)j * m + i0;
                for (int i = 0; i < ni; i++)
                    c[i] += ((a0[i] * b0 + a1[i] * b1) + (a2[i] * b2 + a3[i] * b3));
            }
        }
        for (; l0 < k; l0++) {
            const float *a0 = A + (size_t)l0 * m + i0;
            for (int j = 0; j < n; j++) {
                const float b0 = B[(size_t)l0 * n + j];
                float *c = C + (size_t)j * m + i0;
                for (int i = 0; i < ni; i++) c[i] += a0[i] * b0;
            }
        }
    }
}

/* ------------------------------------------------------------ reductions ----- */
/* cuda/matrix.cu:642-687 (intended result): out[j] = sum_i A[i,j] */
void oracle_sum_cols(const float *A, int rows, int cols, float *out) {
#pragma omp parallel for schedule(static)
    for (int j = 0; j < cols; j++) {
        const float *a = A + (size_t)j * rows;
        float p[8] = {0};
        int i = 0;
        for (; i + 8 <= rows; i += 8)
            for (int v = 0; v < 8; v++) p[v] += a[i + v];
        float s = ((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7]));
        for (; i < rows; i++) s += a[i];
        out[j] = s;
    }
}

/* cuda/matrix.cu:689-735 (intended result): out[i] = sum_j A[i,j] */
void oracle_sum_rows(const float *A, int rows, int cols, float *out) {
    /* fixed two-level order: 64 column blocks, then combine in block order */
    enum { NB = 64 };
    float *part = (float *)calloc((size_t)NB * rows, sizeof(float));
    const int per = (cols + NB - 1) / NB;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < NB; b++) {
        float *p = part + (size_t)b * rows;
        const int j1 = (b + 1) * per < cols ? (b + 1) * per : cols;
        for (int j = b * per; j < j1; j++) {
            const float *a = A + (size_t)j * rows;
            for (int i = 0; i < rows; i++) p[i] += a[i];
        }
    }
    for (int i = 0; i < rows; i++) out[i] = 0.f;
    for (int b = 0; b < NB; b++)
        for (int i = 0; i < rows; i++) out[i] += part[(size_t)b * rows + i];
    free(part);
}

/* What reduce2d<128> actually returned on the reference author's machine, as pinned
 * by Wtest.bin/Htest.bin: thread t holds the partial of rows == t (mod
 * 128) (cuda/matrix.cu:646-655); the __syncthreads stage folds 128->64
 * (cuda/matrix.cu:670-674); the unsynchronised warp tail (cuda/matrix.cu:676-683) was
 * compiled with its loads hoisted, so thread 0 returns
 * s[0]+s[32]+s[16]+s[8]+s[4]+s[2]+s[1] with s[t] = part[t]+part[t+64]. */
void oracle_sum_cols_refcompat(const float *A, int rows, int cols, float *out) {
    static const int tail[7] = {0, 32, 16, 8, 4, 2, 1};
#pragma omp parallel for schedule(static)
    for (int j = 0; j < cols; j++) {
        const float *a = A + (size_t)j * rows;
        float part[128];
        for (int t = 0; t < 128; t++) part[t] = 0.f;
        /* thread t: i = t; while (i < rows-128) part += a[i] + a[i+128], i += 256 */
        for (int t = 0; t < 128; t++) {
            int i = t;
            while (i < rows - 128) { part[t] += a[i] + a[i + 128]; i += 256; }
            if (i < rows) part[t] += a[i];
        }
        float s[64];
        for (int t = 0; t < 64; t++) s[t] = part[t] + part[t + 64];
        float r = s[tail[0]];
        for (int q = 1; q < 7; q++) r += s[tail[q]];
        out[j] = r;
    }
}

/* cuda/matrix.cu:592: x*(logf(x)-logf(y)) - x + y ; accumulated in fp64 here. */
double oracle_kl_div(const float *X, const float *Y, size_t n) {
    double s = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : s)
    for (size_t i = 0; i < n; i++) {
        const double x = X[i], y = Y[i];
        s += x * (log(x) - log(y)) - x + y;
    }
    return s;
}

/* cuda/matrix.cu:517-518: sum |a-b| and sum |a| */
double oracle_rel_l1(const float *X, const float *Y, size_t n) {
    double d = 0.0, a = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : d, a)
    for (size_t i = 0; i < n; i++) {
        d += fabs((double)X[i] - (double)Y[i]);
        a += fabs((double)X[i]);
    }
    return d / a;
}

/* ------------------------------------------------------------- half-steps ---- */
/* Z = X ./ max(W*H, EPS)   (cuda/nmf.cu:125-131 and 155-161) */
static void oracle_quotient(const float *W, const float *H, const float *X, int M, int N, int K,
                            float *Z) {
    oracle_sgemm_nn(M, N, K, W, H, Z);
    const size_t n = (size_t)M * N;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        float wh = Z[i];
        if (wh < ORACLE_EPS) wh = ORACLE_EPS;   /* set_epsilon, cuda/matrix.cu:185-186 */
        Z[i] = X[i] / wh;                       /* vec_div, cuda/matrix.cu:149 */
    }
}

/* cuda/nmf.cu:118-146 */
void oracle_update_h(float *W, float *H, const float *X, int M, int N, int K, float *Z,
                     float *WtZ, float *sumW, int mode) {
    oracle_quotient(W, H, X, M, N, K, Z);
    if (mode == ORACLE_MODE_REFCOMPAT) oracle_sum_cols_refcompat(W, M, K, sumW);
    else                               oracle_sum_cols(W, M, K, sumW);     /* nmf.cu:134 */
    oracle_set_epsilon(sumW, (size_t)K);                                    /* nmf.cu:135 */
    oracle_sgemm_tn(K, N, M, W, Z, WtZ);                                    /* nmf.cu:138 */
#pragma omp parallel for schedule(static)
    for (int j = 0; j < N; j++)
        for (int i = 0; i < K; i++) {
            const size_t ix = (size_t)j * K + i;
            const float q = WtZ[ix] / sumW[i];     /* col_div, cuda/matrix.cu:244-250 */
            H[ix] = H[ix] * q;                     /* vec_mul, cuda/matrix.cu:174-180 */
        }
}

/* cuda/nmf.cu:148-176 */
void oracle_update_w(float *W, float *H, const float *X, int M, int N, int K, float *Z,
                     float *ZHt, float *sumH, int mode) {
    oracle_quotient(W, H, X, M, N, K, Z);
    oracle_sum_rows(H, K, N, sumH);                                         /* nmf.cu:164 */
    oracle_set_epsilon(sumH, (size_t)K);                                    /* nmf.cu:165 */
    oracle_sgemm_nt(M, K, N, Z, H, ZHt);                                    /* nmf.cu:168 */
    /* row_div launches blockDim = rows_padded (cuda/matrix.cu:215-217): an invalid
     * launch when rows_padded > 1024, never reported, so the reference skips it. */
    const int Mp = (M % 32) ? M + (32 - M % 32) : M;
    const int apply_row_div = !(mode == ORACLE_MODE_REFCOMPAT && Mp > 1024);
#pragma omp parallel for schedule(static)
    for (int j = 0; j < K; j++) {
        const float s = sumH[j];
        for (int i = 0; i < M; i++) {
            const size_t ix = (size_t)j * M + i;
            float q = ZHt[ix];
            if (apply_row_div) q = q / s;          /* row_div, cuda/matrix.cu:220-224 */
            W[ix] = W[ix] * q;                     /* vec_mul */
        }
    }
}

/* cuda/nmf.cu:76-116 + README.md:40-54 */
int oracle_update_div(float *W, float *H, const float *X_in, int M, int N, int K, float thresh,
                      int max_iter, int iter_check, int mode, double *kl_trace, int kl_cap,
                      int *n_kl) {
    const size_t mn = (size_t)M * N;
    float *X = (float *)malloc(mn * sizeof(float));
    float *Z = (float *)malloc(mn * sizeof(float));
    float *WtZ = (float *)malloc((size_t)K * N * sizeof(float));
    float *ZHt = (float *)malloc((size_t)M * K * sizeof(float));
    float *sumW = (float *)malloc((size_t)K * sizeof(float));
    float *sumH = (float *)malloc((size_t)K * sizeof(float));
    if (!X || !Z || !WtZ || !ZHt || !sumW || !sumH) { fprintf(stderr, "oracle: out of memory\n"); exit(1); }
    memcpy(X, X_in, mn * sizeof(float));
    /* read_matrix clamps every input to >= EPS (cuda/nmf.cu:210-211) */
    oracle_set_epsilon(X, mn);
    oracle_set_epsilon(W, (size_t)M * K);
    oracle_set_epsilon(H, (size_t)K * N);

    int nk = 0;
    double prev = 0.0;
    const int want_kl = (kl_trace != NULL && kl_cap > 0) || (thresh > 0.f && iter_check > 0);
    if (want_kl) {
        oracle_sgemm_nn(M, N, K, W, H, Z);
        oracle_set_epsilon(Z, mn);
        prev = oracle_kl_div(X, Z, mn);
        if (kl_trace && nk < kl_cap) kl_trace[nk] = prev;
        nk++;
    }
    int it = 0;
    while (it < max_iter) {
        oracle_update_h(W, H, X, M, N, K, Z, WtZ, sumW, mode);
        oracle_update_w(W, H, X, M, N, K, Z, ZHt, sumH, mode);
        it++;
        if (want_kl && iter_check > 0 && (it % iter_check) == 0) {
            oracle_sgemm_nn(M, N, K, W, H, Z);
            oracle_set_epsilon(Z, mn);
            const double cur = oracle_kl_div(X, Z, mn);
            if (kl_trace && nk < kl_cap) kl_trace[nk] = cur;
            nk++;
            /* README.md:51: ratio of cost function change to cost function value */
            if (thresh > 0.f && (prev - cur) / prev < (double)thresh) { prev = cur; break; }
            prev = cur;
        }
    }
    if (n_kl) *n_kl = nk;
    free(X); free(Z); free(WtZ); free(ZHt); free(sumW); free(sumH);
    return it;
}

/* ------------------------------------------------------------------ file I/O -- */
/* cuda/nmf.cu:188-218 (without the device upload / clamp) */
int oracle_read_bin(const char *path, uint32_t *rows, uint32_t *cols, float **data) {
    FILE *fp = fopen(path, "rb");
    if (!fp) return -1;
    uint32_t rc[2];
    if (fread(rc, sizeof(uint32_t), 2, fp) != 2) { fclose(fp); return -2; }
    const size_t n = (size_t)rc[0] * rc[1];
    float *buf = (float *)malloc(n * sizeof(float));
    if (!buf) { fclose(fp); return -3; }
    if (fread(buf, sizeof(float), n, fp) != n) { free(buf); fclose(fp); return -4; }
    fclose(fp);
    *rows = rc[0]; *cols = rc[1]; *data = buf;
    return 0;
}

/* cuda/nmf.cu:220-259 */
int oracle_write_bin(const char *path, uint32_t rows, uint32_t cols, const float *data) {
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    uint32_t rc[2] = {rows, cols};
    const size_t n = (size_t)rows * cols;
    int ok = fwrite(rc, sizeof(uint32_t), 2, fp) == 2 && fwrite(data, sizeof(float), n, fp) == n;
    fclose(fp);
    return ok ? 0 : -2;
}
