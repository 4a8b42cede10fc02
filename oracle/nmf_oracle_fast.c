/*
 * nmf_oracle_fast.c -- the timed CPU baseline of bench.py (TEST INFRASTRUCTURE ONLY, like the rest of oracle/).
 *
 * Same algorithm as oracle_update_div in `spec` mode (cuda/nmf.cu:118-176: update_h then update_w, EPS clamps,
 * IEEE division), arranged for speed instead of for mirroring the reference's call sequence: of the oracle's three
 * SGEMM kernels the rank-1-update one (oracle_sgemm_nn) runs ~8x faster per core than the dot-product one
 * (oracle_sgemm_tn) and ~4x faster than oracle_sgemm_nt on many cores, so W' * Z and Z * H' are computed as
 * (W')(explicitly transposed) * Z and Z * (H')(explicitly transposed) through it; the transposes cost two passes over
 * W and H.  The CPU number printed next to the GPU's is then what this host reaches with the oracle's best kernel,
 * not its slowest.  Checked against oracle_update_div by tolerance (tests/test_oracle_ops.py); never the parity
 * checker itself -- the summation order differs from the pinned oracle's.
 */
#include "nmf_oracle.h"

#include <stdlib.h>
#include <string.h>

/* B(c x r) = A(r x c)', both column-major, 32 x 32 blocks */
static void transpose(const float *restrict A, int r, int c, float *restrict B) {
#pragma omp parallel for schedule(static)
    for (int j0 = 0; j0 < c; j0 += 32)
        for (int i0 = 0; i0 < r; i0 += 32) {
            const int j1 = (j0 + 32 < c) ? j0 + 32 : c, i1 = (i0 + 32 < r) ? i0 + 32 : r;
            for (int j = j0; j < j1; j++)
                for (int i = i0; i < i1; i++) B[(size_t)i * c + j] = A[(size_t)j * r + i];
        }
}

/* Z = X ./ max(W*H, EPS)   (cuda/nmf.cu:125-131, 155-161) */
static void quotient(const float *W, const float *H, const float *X, int M, int N, int K, float *Z) {
    oracle_sgemm_nn(M, N, K, W, H, Z);
    const size_t n = (size_t)M * N;
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; i++) {
        float wh = Z[i];
        if (wh < ORACLE_EPS) wh = ORACLE_EPS;
        Z[i] = X[i] / wh;
    }
}

/* `iters` iterations of update_h, update_w (W, H in place; X read-only, clamped on a private copy, cuda/nmf.cu:211).
 * Returns 0, or -1 when scratch cannot be allocated. */
int oracle_fast_update_div(float *W, float *H, const float *X, int M, int N, int K, int iters) {
    const size_t mn = (size_t)M * N, mk = (size_t)M * K, kn = (size_t)K * N;
    float *Xc = malloc(mn * sizeof(float)), *Z = malloc(mn * sizeof(float));
    float *T = malloc((mk > kn ? mk : kn) * sizeof(float)), *P = malloc((mk > kn ? mk : kn) * sizeof(float));
    float *sum = malloc((size_t)K * sizeof(float));
    if (!Xc || !Z || !T || !P || !sum) { free(Xc); free(Z); free(T); free(P); free(sum); return -1; }
    memcpy(Xc, X, mn * sizeof(float));
    oracle_set_epsilon(Xc, mn);
    oracle_set_epsilon(W, mk);
    oracle_set_epsilon(H, kn);
    for (int it = 0; it < iters; it++) {
        /* update_h: H .*= (W' * Z) ./ colsum(W) */
        quotient(W, H, Xc, M, N, K, Z);
        oracle_sum_cols(W, M, K, sum);
        oracle_set_epsilon(sum, (size_t)K);
        transpose(W, M, K, T);                       /* T = W' (K x M) */
        oracle_sgemm_nn(K, N, M, T, Z, P);           /* P = W' * Z (K x N) */
#pragma omp parallel for schedule(static)
        for (int j = 0; j < N; j++)
            for (int k = 0; k < K; k++) {
                const size_t ix = (size_t)j * K + k;
                H[ix] = H[ix] * (P[ix] / sum[k]);
            }
        /* update_w: W .*= (Z * H') ./ rowsum(H) */
        quotient(W, H, Xc, M, N, K, Z);
        oracle_sum_rows(H, K, N, sum);
        oracle_set_epsilon(sum, (size_t)K);
        transpose(H, K, N, T);                       /* T = H' (N x K) */
        oracle_sgemm_nn(M, K, N, Z, T, P);           /* P = Z * H' (M x K) */
#pragma omp parallel for schedule(static)
        for (int k = 0; k < K; k++)
            for (int i = 0; i < M; i++) {
                const size_t ix = (size_t)k * M + i;
                W[ix] = W[ix] * (P[ix] / sum[k]);
            }
    }
    free(Xc); free(Z); free(T); free(P); free(sum);
    return 0;
}
