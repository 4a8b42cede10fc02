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
 *
 * The two long reductions (over M in W' * Z, over N in Z * H') go through sgemm_nn_long below, which sums the reduction
 * dimension in blocks of 512 and adds the block results: one fp32 accumulator run over all 65536 columns of BASELINE
 * config 3 comes out LOW by 8.0e-7 +- 0.6e-7 relative (a bias of sequential round-to-nearest summation of positive terms,
 * measured over 3000 random sums of that length) while rowsum(H) is summed in 64 blocks and has none, so W shrank by that
 * factor every iteration and H grew by it -- W*H unchanged, the factors 2.3e-4 from the GPU's after 200 iterations at
 * config 3, all of it one scalar (round 3; against an fp64 evaluation of the same iteration, 256 x 65536 x 32, 50
 * iterations: scale of W -4.3e-5 with the single accumulator, -1e-7 for the pinned oracle's 8-lane partial sums).
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

/* C(m x n) = A(m x k) * B(k x n) like oracle_sgemm_nn (rank-1 updates over 512-row panels, four columns at a time), with
 * the reduction dimension summed in blocks of LB: block sums in t*, added to the running result c* once per block */
static void sgemm_nn_long(int m, int n, int k, const float *restrict A, const float *restrict B, float *restrict C) {
    enum { IB = 1024, LB = 512 };
#pragma omp parallel for schedule(dynamic, 1)
    for (int j0 = 0; j0 < n; j0 += 4) {
        const int nj = (n - j0 < 4) ? (n - j0) : 4;
        for (int i0 = 0; i0 < m; i0 += IB) {
            const int ni = (m - i0 < IB) ? (m - i0) : IB;
            float *c0 = C + (size_t)(j0 + 0) * m + i0;
            float *c1 = C + (size_t)(j0 + (nj > 1 ? 1 : 0)) * m + i0;
            float *c2 = C + (size_t)(j0 + (nj > 2 ? 2 : 0)) * m + i0;
            float *c3 = C + (size_t)(j0 + (nj > 3 ? 3 : 0)) * m + i0;
            float t0[IB], t1[IB], t2[IB], t3[IB];
            for (int l0 = 0; l0 < k; l0 += LB) {
                const int l1 = (l0 + LB < k) ? l0 + LB : k;
                memset(t0, 0, sizeof(float) * ni); memset(t1, 0, sizeof(float) * ni);
                memset(t2, 0, sizeof(float) * ni); memset(t3, 0, sizeof(float) * ni);
                for (int l = l0; l < l1; l++) {
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
                /* block sums into the result (first block: a copy); highest alias last so that nj < 4 duplicates resolve to column j0 */
                if (l0 == 0) {
                    if (nj > 3) memcpy(c3, t3, sizeof(float) * ni);
                    if (nj > 2) memcpy(c2, t2, sizeof(float) * ni);
                    if (nj > 1) memcpy(c1, t1, sizeof(float) * ni);
                    memcpy(c0, t0, sizeof(float) * ni);
                } else {
                    if (nj > 3) for (int i = 0; i < ni; i++) c3[i] += t3[i];
                    if (nj > 2) for (int i = 0; i < ni; i++) c2[i] += t2[i];
                    if (nj > 1) for (int i = 0; i < ni; i++) c1[i] += t1[i];
                    for (int i = 0; i < ni; i++) c0[i] += t0[i];
                }
            }
        }
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
        sgemm_nn_long(K, N, M, T, Z, P);             /* P = W' * Z (K x N), reduction over M */
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
        sgemm_nn_long(M, K, N, Z, T, P);             /* P = Z * H' (M x K), reduction over N */
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
