// nmf_device.h -- device-side helpers shared by the kernel translation units (gfx950, wave64).
#ifndef NMF_DEVICE_H
#define NMF_DEVICE_H
#include "nmf_kernels.h"

#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

namespace nmf {

// Kernels that need more than 64 KiB of dynamic LDS must opt in once per (kernel, device).
inline hipError_t ensure_dynamic_lds(const void *fn, size_t bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void *, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({fn, dev})) return hipSuccess;
    e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) done.insert({fn, dev});
    return e;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define NMF_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ constexpr int rho(int r) { return (r & 3) + 8 * (r >> 2); }

// set_epsilon semantics (cuda/matrix.cu:185-186): a clamp, NaN passes through.
__device__ __forceinline__ float clamp_eps(float v) { return (v < kEps) ? kEps : v; }
// operands in [EPS, 2^60] (or a zero numerator) never trigger the range scaling of the IEEE division sequence
constexpr float kDivSafeMax = 1152921504606846976.0f;   // 2^60

// v_log_f32: log2 to 1 ulp in one quarter-rate instruction.  The check kernels sum x * log2(y) with it and the factor ln 2 is
// applied once, in fp64, by check_compose_kernel: a one-ulp error per term, even if it were all bias, moves sum x log y by
// 6e-8 of itself, i.e. the KL value by < 1e-6 relative while KL >= 6 % of sum |x log y| (every run of the tests: 15 %..300 %);
// libm-accurate logf costs ~20 VALU instructions per element on a datapath the f32 MFMA shares.
__device__ __forceinline__ float log2_hw(float y) { return __builtin_amdgcn_logf(y); }

// 64-lane sum, result valid in lane 0
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// one workgroup (4 waves) -> out3[0..2], fixed order
__device__ __forceinline__ void block_reduce3(double v0, double v1, double v2, double *out3, int tid) {
    __shared__ double red[3][4];
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = v0; red[1][tid >> 6] = v1; red[2][tid >> 6] = v2; }
    __syncthreads();
    if (tid == 0) {
        out3[0] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
        out3[1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
        out3[2] = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    }
}

// LDS image of one streamed chunk: Vl[k][p], p = 0..31 within the chunk, row stride 33 floats.
//   product-1 A operand  Vl[(2s+h)*33 + c]          : 32 consecutive banks            -> conflict-free
//   product-2 A operand  Vl[(32t+c)*33 + rho(r)+4h] : stride 33 (odd) across 32 lanes -> conflict-free
constexpr int kLdv = 33;

// depth of the register ring that carries MFMA A operands from LDS (loaded kRing MFMAs ahead of their use)
constexpr int kRing = 8;
constexpr int kXtLd = 36;                       // X patch row stride in floats (16-B aligned, b128 conflict-free)
constexpr int kXtFloats = 32 * kXtLd;           // per wave (32-column kernel)

// LDS pointer with its address space spelled out: a volatile load through a generic pointer is not
// rewritten by address-space inference and would become flat_load + 64-bit address arithmetic.
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ float lds_ld(const lds_float *p) { return *reinterpret_cast<const volatile lds_float *>(p); }

// ---- shared by the two 16-column kernel families (nmf_fused16_impl.h, nmf_split16_impl.h), K = 16 KT
typedef const __attribute__((address_space(1))) char *global_bytes;
#define NMF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// k index of product-1 step s in lane group kq = l >> 4: whole blocks of 64 first, 64 (s >> 4) + 16 kq + (s & 15) -- per-lane
// contiguous runs of 16 (16-B loads of the owned factor) and, with 33-float LDS rows, 32 distinct banks per half-wave -- then,
// where K is not a multiple of 64, one remainder block of R = (K % 64) / 4 = 4, 8 or 12 steps, in one of two arrangements:
//   IL = false: runs of R per lane group, 64 (K / 64) + R pi(kq) + s', pi = (0, 2, 1, 3): the two lane groups of a half-wave sit
//               2 R apart (conflict-free LDS reads at R = 8, 2-way on half of the lanes at R = 4, 12), 16-B loads of the owned factor;
//   IL = true:  interleaved over the lane groups, 64 (K / 64) + 4 s' + kq: 2-way conflicts on those ds_reads (4 LDS cycles instead of
//               2 next to a 32-cycle MFMA), and in exchange the last steps cover the TOP k indices -- where the caller's K is not a
//               multiple of 16 the last one to three steps hold zero padding only and can be skipped at run time
//               (FusedArgs::p1_trim): product 1 then runs at a granularity of 4 in K while product 2 (16 x 16 tiles) keeps 16.
// NF = product-1 steps in the whole blocks that use the run map (a multiple of 16), RR = steps of the remainder block.
// k16_kconst: the part of the k index that does not depend on the lane ...
template <int NF, bool IL> __host__ __device__ constexpr int k16_kconst(int s) {
    return s < NF ? 64 * (s >> 4) + (s & 15) : 4 * NF + (IL ? 4 : 1) * (s - NF);
}
// ... whether the step lies in the remainder block ...
template <int NF> __host__ __device__ constexpr bool k16_in_rem(int s) { return s >= NF; }
// ... and the lane part of a remainder step (whole blocks: 16 kq)
template <int RR, bool IL> __device__ __forceinline__ int k16_rem_lane(int kq) {
    return IL ? kq : RR * (((kq & 1) << 1) | (kq >> 1));
}

// DIV = 0: correctly rounded IEEE division (hipcc's expansion of `/`, 11 VALU);
// DIV = 1: reciprocal refined to <= 1 ulp (rcp, 2 fma, mul, 2 fma; no scaling: y >= EPS is normal here)
template <int DIV>
__device__ __forceinline__ float quotient(float x, float y) {
    if (DIV == 0) return x / y;
    float r = __builtin_amdgcn_rcpf(y);
    r = __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
    const float q = x * r;
    return __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
}

// Eight quotients z[r] = x[r] / max(s[r], EPS) of one lane (one chunk of the 16-column kernel).
// DIV = 1: quotient<1> each, unconditionally.  DIV = 0: correctly rounded.  hipcc expands `/` into
//     ys = div_scale(y), xs = div_scale(x), r0 = rcp(ys), r = fma(fma(-ys, r0, 1), r0, r0), q = xs * r,
//     q = fma(fma(-ys, q, xs), r, q), q = div_fmas(fma(-ys, q, xs), r, q), div_fixup(q, y, x)      (11 VALU + 2 for the clamp)
// whose div_scale / div_fmas / div_fixup only act when an operand or the quotient leaves the normal range, and whose
// last correction never changes the result there: quotient<1> (6 VALU) returns the same bits for EVERY pair of fp32
// significands -- all 2^46 enumerated on the device, and v_rcp_f32 checked exponent-invariant (tools/divide_exhaustive.py,
// profiles/r01_divide_exhaustive.log) -- hence for every x = 0 or x, y in [EPS, 2^60], where operands, quotient and
// remainders stay normal and every step is exponent-invariant.  X is range-checked once at upload (in_range); the
// denominators per chunk with one integer max over the lane's 8 raw dot products (NaN and negative bit patterns
// compare high).  A wave with everything in range takes quotient<1> behind a one-instruction clamp (v_max_f32 equals
// `s < EPS ? EPS : s` for the non-NaN values that pass the guard); any other wave runs the full sequence.  The f32 MFMA
// shares the VALU datapath (profiles/r01_pmc_summary.md): every VALU instruction saved here is MFMA issue time.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int DIV>
__device__ __forceinline__ void quotient8(const float (&x)[8], const f32x4 &s0, const f32x4 &s1, float (&z)[8], bool in_range) {
    if (DIV == 1) {
#pragma unroll
        for (int r = 0; r < 8; ++r) z[r] = quotient<1>(x[r], clamp_eps(r < 4 ? s0[r] : s1[r - 4]));
        return;
    }
    unsigned m = __float_as_uint(s0[0]);
#pragma unroll
    for (int r = 1; r < 8; ++r) { const unsigned b = __float_as_uint(r < 4 ? s0[r] : s1[r - 4]); m = b > m ? b : m; }
    const bool fast = in_range && __builtin_amdgcn_ballot_w64(m > __float_as_uint(kDivSafeMax)) == 0;
    if (fast) {
        // quotient<1>, stage by stage over the eight operands so that no instruction waits on its predecessor
        const float eps = kEps;
        float y[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(y[r]) : "v"(r < 4 ? s0[r] : s1[r - 4]), "v"(eps));
        f32x2 yy[4], xx[4], rc[4], q[4], e[4];
        const f32x2 one = {1.0f, 1.0f};
#pragma unroll
        for (int i = 0; i < 4; ++i) { yy[i] = f32x2{y[2 * i], y[2 * i + 1]}; xx[i] = f32x2{x[2 * i], x[2 * i + 1]}; }
#pragma unroll
        for (int i = 0; i < 4; ++i) rc[i] = f32x2{__builtin_amdgcn_rcpf(yy[i].x), __builtin_amdgcn_rcpf(yy[i].y)};
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_elementwise_fma(-yy[i], rc[i], one);
#pragma unroll
        for (int i = 0; i < 4; ++i) rc[i] = __builtin_elementwise_fma(e[i], rc[i], rc[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = xx[i] * rc[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_elementwise_fma(-yy[i], q[i], xx[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = __builtin_elementwise_fma(e[i], rc[i], q[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) { z[2 * i] = q[i].x; z[2 * i + 1] = q[i].y; }
    } else {
#pragma unroll
        for (int r = 0; r < 8; ++r) z[r] = x[r] / clamp_eps(r < 4 ? s0[r] : s1[r - 4]);
    }
}

// Variants (NMF_FUSED_VARIANT), for K <= 256: unset = production choice (16-column kernel at two workgroups per CU for
// K = 64/128/256, v3 for K = 32); 3 = v3 (32-column kernel) everywhere; 1 = first chunk-serial kernel (64-bit addressing,
// ablation probes).
inline int fused_variant() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NMF_FUSED_VARIANT"); v = (e && (e[0] == '1' || e[0] == '3')) ? (e[0] - '0') : 0; }   // 0 = automatic choice
    return v;
}
// grid of the grid-stride elementwise kernels
inline unsigned ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

}  // namespace nmf
#endif
