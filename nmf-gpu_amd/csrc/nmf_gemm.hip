// nmf_gemm.hip -- the reference's three SGEMM operators (cuda/matrix.cu:97-125) as fp32 MFMA kernels; also the
// iteration path for K > 512.
#include "nmf_device.h"

namespace nmf {

// =====================================================================================
// Unfused operators (one per reference operator; also the K > 256 fallback)
// =====================================================================================
// Generic fp32 MFMA GEMM, 128 x 128 x 16 tiles, 4 waves (2 x 2), each wave 64 x 64 = 2 x 2 MFMA
// tiles.  The MFMA is issued "transposed" (B-side value as the A operand) so that the lane index
// of the result runs along the rows of C: stores are 128-byte coalesced in column-major C.
//   A(i,l) = A[i*sai + l*sal],  B(l,j) = B[l*sbl + j*sbj],  C(i,j) = C[i + j*ldc]
constexpr int kGemmLd = 129;
template <bool A_LCONTIG, bool B_LCONTIG>
__global__ __launch_bounds__(256) void gemm_kernel(int m, int n, int k_total, const float *__restrict__ A, long sai, long sal,
                                                   const float *__restrict__ B, long sbl, long sbj, float *__restrict__ C, long ldc,
                                                   int k_per_split, size_t slab) {
    // split-K: blockIdx.z owns reduction range [z*k_per_split, ...) and writes its own slab of C
    const int k_begin = blockIdx.z * k_per_split;
    const int k = (k_begin + k_per_split < k_total) ? (k_begin + k_per_split) : k_total;   // exclusive end
    C += (size_t)blockIdx.z * slab;
    __shared__ float As[16 * kGemmLd];
    __shared__ float Bs[16 * kGemmLd];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int i_base = blockIdx.x * 128, j_base = blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    float ra[8], rb[8];
    auto fetch = [&](int l0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * 256;
            const int la = A_LCONTIG ? (e & 15) : (e >> 7), ia = A_LCONTIG ? (e >> 4) : (e & 127);
            const int gi = i_base + ia, gl = l0 + la;
            ra[q] = (gi < m && gl < k) ? A[(size_t)gi * sai + (size_t)gl * sal] : 0.f;
            const int lb = B_LCONTIG ? (e & 15) : (e >> 7), jb = B_LCONTIG ? (e >> 4) : (e & 127);
            const int gj = j_base + jb, gl2 = l0 + lb;
            rb[q] = (gj < n && gl2 < k) ? B[(size_t)gl2 * sbl + (size_t)gj * sbj] : 0.f;
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int e = tid + q * 256;
            const int la = A_LCONTIG ? (e & 15) : (e >> 7), ia = A_LCONTIG ? (e >> 4) : (e & 127);
            As[la * kGemmLd + ia] = ra[q];
            const int lb = B_LCONTIG ? (e & 15) : (e >> 7), jb = B_LCONTIG ? (e >> 4) : (e & 127);
            Bs[lb * kGemmLd + jb] = rb[q];
        }
    };
    fetch(k_begin);
    for (int l0 = k_begin; l0 < k; l0 += 16) {
        commit();
        __syncthreads();
        if (l0 + 16 < k) fetch(l0 + 16);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) av[t] = As[(2 * kk + h) * kGemmLd + wm * 64 + t * 32 + c];
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[u] = Bs[(2 * kk + h) * kGemmLd + wn * 64 + u * 32 + c];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = NMF_MFMA(bv[u], av[t], acc[t][u]);   // D[j-off][i-off]: lane runs along i
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gi = i_base + wm * 64 + t * 32 + c;
                const int gj = j_base + wn * 64 + u * 32 + rho(r) + 4 * h;
                if (gi < m && gj < n) C[(size_t)gi + (size_t)gj * ldc] = acc[t][u][r];
            }
}

// Fast path of the generic GEMM for tile-aligned problems (m, n multiples of 128, k of 16, 16-B aligned operands):
// 16-B global loads (2 per operand per thread per k-tile instead of 8 scalar ones with bounds tests), double-buffered
// LDS with one barrier per k-tile, no bounds arithmetic.  Same tiling, lane maps and summation order as gemm_kernel.
constexpr int kGemmLdF = 132;   // LDS row stride: 16-B aligned rows for ds_write_b128
template <bool A_LCONTIG, bool B_LCONTIG>
__global__ __launch_bounds__(256) void gemm_fast_kernel(int m, int n, int k_total, const float *__restrict__ A, long sai, long sal,
                                                        const float *__restrict__ B, long sbl, long sbj, float *__restrict__ C, long ldc,
                                                        int k_per_split, size_t slab) {
    __shared__ __attribute__((aligned(16))) float As[2][16 * kGemmLdF];
    __shared__ __attribute__((aligned(16))) float Bs[2][16 * kGemmLdF];
    const int k_begin = blockIdx.z * k_per_split;
    const int k_end = (k_begin + k_per_split < k_total) ? (k_begin + k_per_split) : k_total;
    C += (size_t)blockIdx.z * slab;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int i_base = blockIdx.x * 128, j_base = blockIdx.y * 128;
    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    // operand contiguous along the tile's outer index (i or j): piece o4 = tid & 31 (4 consecutive i), row l = (tid >> 5) + 8 q
    // operand contiguous along l: piece l4 = tid & 3 (4 consecutive l), column o = (tid >> 2) + 64 q
    f32x4 ra[2], rb[2];
    auto fetch = [&](int l0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!A_LCONTIG) ra[q] = *reinterpret_cast<const f32x4 *>(A + (size_t)(i_base + 4 * (tid & 31)) * sai + (size_t)(l0 + (tid >> 5) + 8 * q) * sal);
            else            ra[q] = *reinterpret_cast<const f32x4 *>(A + (size_t)(i_base + (tid >> 2) + 64 * q) * sai + (size_t)(l0 + 4 * (tid & 3)) * sal);
            if (!B_LCONTIG) rb[q] = *reinterpret_cast<const f32x4 *>(B + (size_t)(j_base + 4 * (tid & 31)) * sbj + (size_t)(l0 + (tid >> 5) + 8 * q) * sbl);
            else            rb[q] = *reinterpret_cast<const f32x4 *>(B + (size_t)(j_base + (tid >> 2) + 64 * q) * sbj + (size_t)(l0 + 4 * (tid & 3)) * sbl);
        }
    };
    auto commit = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            if (!A_LCONTIG) *reinterpret_cast<f32x4 *>(&As[buf][((tid >> 5) + 8 * q) * kGemmLdF + 4 * (tid & 31)]) = ra[q];
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[buf][(4 * (tid & 3) + e) * kGemmLdF + (tid >> 2) + 64 * q] = ra[q][e];
            }
            if (!B_LCONTIG) *reinterpret_cast<f32x4 *>(&Bs[buf][((tid >> 5) + 8 * q) * kGemmLdF + 4 * (tid & 31)]) = rb[q];
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) Bs[buf][(4 * (tid & 3) + e) * kGemmLdF + (tid >> 2) + 64 * q] = rb[q][e];
            }
        }
    };
    fetch(k_begin);
    commit(0);
    __syncthreads();
    int buf = 0;
    for (int l0 = k_begin; l0 < k_end; l0 += 16) {
        const bool more = l0 + 16 < k_end;
        if (more) fetch(l0 + 16);
        const float *__restrict__ as = As[buf] + h * kGemmLdF + wm * 64 + c;
        const float *__restrict__ bs = Bs[buf] + h * kGemmLdF + wn * 64 + c;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            float av[2], bv[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) av[t] = as[2 * kk * kGemmLdF + t * 32];
#pragma unroll
            for (int u = 0; u < 2; ++u) bv[u] = bs[2 * kk * kGemmLdF + u * 32];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) acc[t][u] = NMF_MFMA(bv[u], av[t], acc[t][u]);
        }
        if (more) commit(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(size_t)(i_base + wm * 64 + t * 32 + c) + (size_t)(j_base + wn * 64 + u * 32 + rho(r) + 4 * h) * ldc] = acc[t][u][r];
}

hipError_t launch_gemm(GemmKind kind, int m, int n, int k, const float *A, long lda, const float *B, long ldb, float *C, long ldc,
                       hipStream_t stream, float *workspace, size_t workspace_floats) {
    if (m <= 0 || n <= 0 || k <= 0) return hipErrorInvalidValue;
    // the NMF product W * H (tall A, K <= 512, everything contiguous): the fused kernels' product-1 engine, B in registers
    static const bool generic_only = getenv("NMF_GEMM_GENERIC") != nullptr;   // A/B switch (tools/gemm_bench.py)
    if (!generic_only && kind == GEMM_NN && m >= 1024 && gemm_nn16_eligible(m, n, k, lda, ldb, ldc) &&
        (reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B) | reinterpret_cast<uintptr_t>(C)) % 16 == 0)
        return launch_gemm_nn16(A, B, C, m, n, k, stream);
    const int tiles = ((m + 127) / 128) * ((n + 127) / 128);
    // small output, long reduction (Z*H' of the W-step): split K over workgroups into slabs, then sum them in order
    int nsplit = 1;
    if (workspace && tiles < 256 && ldc == m) {
        nsplit = (512 + tiles - 1) / tiles;
        const int max_by_k = k / 256 > 0 ? k / 256 : 1;
        if (nsplit > max_by_k) nsplit = max_by_k;
        const size_t per = (size_t)m * n;
        if ((size_t)nsplit * per > workspace_floats) nsplit = (int)(workspace_floats / per);
        if (nsplit < 2) nsplit = 1;
    }
    int kper = (k + nsplit - 1) / nsplit;
    kper = (kper + 15) & ~15;
    nsplit = (k + kper - 1) / kper;
    const dim3 grid((m + 127) / 128, (n + 127) / 128, nsplit), block(256);
    float *out = nsplit > 1 ? workspace : C;
    const size_t slab = nsplit > 1 ? (size_t)m * n : 0;
    const long ldo = nsplit > 1 ? (long)m : ldc;
    const bool aligned = (m % 128 == 0) && (n % 128 == 0) && (k % 16 == 0) && (kper % 16 == 0) && (lda % 4 == 0) && (ldb % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(A) | reinterpret_cast<uintptr_t>(B)) % 16 == 0);
#define NMF_GEMM(AL_, BL_, ...)                                                                                             \
    do {                                                                                                                    \
        if (aligned) hipLaunchKernelGGL((gemm_fast_kernel<AL_, BL_>), grid, block, 0, stream, __VA_ARGS__);                 \
        else         hipLaunchKernelGGL((gemm_kernel<AL_, BL_>), grid, block, 0, stream, __VA_ARGS__);                      \
    } while (0)
    switch (kind) {
        case GEMM_NN:   // A(i,l) = A[i + l*lda]; B(l,j) = B[l + j*ldb]
            NMF_GEMM(false, true, m, n, k, A, 1L, lda, B, 1L, ldb, out, ldo, kper, slab);
            break;
        case GEMM_TN:   // A stored (k x m): A(i,l) = A[l + i*lda]
            NMF_GEMM(true, true, m, n, k, A, lda, 1L, B, 1L, ldb, out, ldo, kper, slab);
            break;
        case GEMM_NT:   // B stored (n x k): B(l,j) = B[j + l*ldb]
            NMF_GEMM(false, false, m, n, k, A, 1L, lda, B, ldb, 1L, out, ldo, kper, slab);
            break;
    }
#undef NMF_GEMM
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || nsplit == 1) return e;
    return launch_sum_partials(C, workspace, nsplit, (size_t)m * n, stream);
}

}  // namespace nmf
