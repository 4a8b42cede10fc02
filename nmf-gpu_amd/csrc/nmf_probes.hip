// nmf_probes.hip -- diagnostics only (never on the product path): the micro-probes behind profiles/r01_pmc_summary.md
// and the division census / exhaustive comparison behind DESIGN.md 4.1c.  Reached through nmf_solver_time_piece(which >= 1000).
#ifndef NMF_DIAGNOSTICS
#define NMF_DIAGNOSTICS 1
#endif
#include "nmf_device.h"

namespace nmf {

// Micro-probe: which instruction kinds does a lone wave per SIMD overlap with a running f32 MFMA?
// KIND 0 = v_add_f32 (VALU), 1 = ds_read_b32 (LDS), 2 = s_add_u32 (SALU), 3 = global_load_dword (VMEM),
// 4 = v_accvgpr_read (VALU move), 5 = ds_write_b32.  NV instructions of that kind after every MFMA.
template <int NV, int KIND>
__global__ __launch_bounds__(256, 1) void mfma_mix_probe_kernel(float *out, int iters) {
    __shared__ float lds[1024];
    f32x16 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    lds[threadIdx.x] = 1.0f;
    __syncthreads();
    float a = (float)threadIdx.x, b = 1.0f;
    float d[4] = {1.f, 2.f, 3.f, 4.f};
    unsigned sx = 0;
    const unsigned laddr = (threadIdx.x & 255) * 4;
    const float *gp = out + 65536 + threadIdx.x;
    const float *gp4 = out + 65536 + 4 * threadIdx.x;
    const unsigned laddr4 = (threadIdx.x & 63) * 16;
    f32x4 d4[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 64; ++i) {
            acc[i & 7] = NMF_MFMA(a, b, acc[i & 7]);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(d[v & 3]));
                if (KIND == 1) asm volatile("ds_read_b32 %0, %1" : "=v"(d[v & 3]) : "v"(laddr));
                if (KIND == 2) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sx));
                if (KIND == 3) asm volatile("global_load_dword %0, %1, off" : "=v"(d[v & 3]) : "v"(gp));
                if (KIND == 4) asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(d[v & 3]) : "a"(acc[7][v & 3]));
                if (KIND == 5) asm volatile("ds_write_b32 %0, %1" :: "v"(laddr), "v"(d[v & 3]));
                if (KIND == 6) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d4[v & 3]) : "v"(gp4));
                if (KIND == 7) asm volatile("ds_write_b128 %0, %1" :: "v"(laddr4), "v"(d4[v & 3]));
            }
            if ((KIND == 1 || KIND == 3 || KIND == 5 || KIND == 6 || KIND == 7) && (i & 7) == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)");
    float sum = d[0] + d[1] + d[2] + d[3] + (float)sx + d4[0][0] + d4[1][1] + d4[2][2] + d4[3][3];
#pragma unroll
    for (int t = 0; t < 8; ++t) sum += acc[t][0];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}
hipError_t launch_mfma_valu_probe(int nv, int kind, float *out, int iters, hipStream_t stream) {
#define NMF_MP2(NV_, K_) if (nv == NV_ && kind == K_) { hipLaunchKernelGGL((mfma_mix_probe_kernel<NV_, K_>), dim3(256), dim3(256), 0, stream, out, iters); return hipGetLastError(); }
#define NMF_MP(K_) NMF_MP2(0, K_) NMF_MP2(1, K_) NMF_MP2(2, K_) NMF_MP2(4, K_)
    NMF_MP(0) NMF_MP(1) NMF_MP(2) NMF_MP(3) NMF_MP(4) NMF_MP(5) NMF_MP(6) NMF_MP(7)
#undef NMF_MP
#undef NMF_MP2
    return hipErrorInvalidValue;
}

// Micro-probe 2: two waves per SIMD (512-thread workgroup).  Waves 0-3 issue only f32 MFMAs, waves 4-7
// only VALU (MODE 1), only LDS reads (MODE 2) or nothing (MODE 0).  Does the partner's work slow the MFMAs?
template <int MODE>
__global__ __launch_bounds__(512, 2) void mfma_partner_probe_kernel(float *out, int iters) {
    __shared__ float lds[1024];
    lds[threadIdx.x & 1023] = 1.0f;
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float sum = 0.f;
    if (wave < 4) {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        float a = (float)threadIdx.x, b = 1.0f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 64; ++i) acc[i & 3] = NMF_MFMA(a, b, acc[i & 3]);
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) sum += acc[t][0];
    } else if (MODE == 1) {
        float d0 = 1.f, d1 = 2.f, d2 = 3.f, d3 = 4.f;
        for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d0)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d1));
                asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d2)); asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(d3));
            }
        }
        sum = d0 + d1 + d2 + d3;
    } else if (MODE == 2) {
        const unsigned laddr = (threadIdx.x & 255) * 4;
        float d0 = 0.f;
        for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
            for (int i = 0; i < 32; ++i) asm volatile("ds_read_b32 %0, %1" : "=v"(d0) : "v"(laddr));
            asm volatile("s_waitcnt lgkmcnt(0)");
        }
        sum = d0;
    }
    out[blockIdx.x * 512 + threadIdx.x] = sum;
}
hipError_t launch_mfma_partner_probe(int mode, float *out, int iters, hipStream_t stream) {
    if (mode == 0) hipLaunchKernelGGL((mfma_partner_probe_kernel<0>), dim3(256), dim3(512), 0, stream, out, iters);
    else if (mode == 1) hipLaunchKernelGGL((mfma_partner_probe_kernel<1>), dim3(256), dim3(512), 0, stream, out, iters);
    else hipLaunchKernelGGL((mfma_partner_probe_kernel<2>), dim3(256), dim3(512), 0, stream, out, iters);
    return hipGetLastError();
}

// Diagnostic: how often does quotient<1> (refined reciprocal) differ from the IEEE quotient on operands of the
// kind the kernel sees (x in [EPS, 2), y = clamped dot products in [EPS, 300))?  counts[0] = mismatches, counts[1] = max ulp distance.
__global__ __launch_bounds__(256) void divide_compare_kernel(unsigned long long *counts, unsigned seed, int per_thread) {
    unsigned st = seed ^ (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    unsigned long long bad = 0, maxulp = 0;
    for (int i = 0; i < per_thread; ++i) {
        st = st * 1664525u + 1013904223u; const unsigned a = st;
        st = st * 1664525u + 1013904223u; const unsigned b = st;
        st = st * 1664525u + 1013904223u; const unsigned m = st >> 28;
        float x = (float)(a >> 8) * (1.0f / 16777216.0f) * 2.0f;
        float y = (float)(b >> 8) * (1.0f / 16777216.0f);
        // spread y over many binades: 2^-52 .. 2^8
        y = ldexpf(y + 0.5f, (int)(m * 4) - 52 + (int)((st >> 20) & 3));
        x = clamp_eps(x); y = clamp_eps(y);
        if (m == 0) x = ldexpf(x, -40);   // small numerators too
        const float q0 = quotient<0>(x, y), q1 = quotient<1>(x, y);
        if (q0 != q1) {
            ++bad;
            const long long d = (long long)__float_as_int(q0) - (long long)__float_as_int(q1);
            const unsigned long long ad = d < 0 ? -d : d;
            if (ad > maxulp) maxulp = ad;
        }
    }
    atomicAdd(&counts[0], bad);
    atomicMax(&counts[1], maxulp);
}
hipError_t launch_divide_compare(unsigned long long *counts, unsigned seed, hipStream_t stream) {
    hipLaunchKernelGGL(divide_compare_kernel, dim3(4096), dim3(256), 0, stream, counts, seed, 1000);
    return hipGetLastError();
}

// Diagnostic: EXHAUSTIVE comparison of quotient<1> with the IEEE quotient over every pair of fp32 significands
// (2^23 x 2^23; x = 1.mx, y = 1.my).  All operations of both sequences are exponent-invariant while operands,
// quotient and remainders stay normal (rcp checked separately below), so zero mismatches here proves the two
// bit-identical for every x, y in [EPS, 2^60].  One launch covers 2^17 denominators (slice of 64).
// counts[0] = mismatches, counts[1] = pairs compared, counts[2] = threads that saw one, counts[3..10] = the first
// mismatching (mx, my) of up to 8 of them.  VARIANT 1 (no refinement of the reciprocal, 4 instructions) fails on
// 47 045 of the 2^46 pairs; VARIANT 2 (x * rcp(y)) is the self-check of the harness (20 % mismatches).
template <int VARIANT>
__global__ __launch_bounds__(256) void divide_exhaustive_kernel(unsigned long long *counts, unsigned slice) {
    const unsigned t = blockIdx.x * 256u + threadIdx.x;            // 2^20 threads
    const unsigned my = (slice << 17) | (t >> 3);
    const unsigned x0 = (t & 7u) << 20;
    const float y = __uint_as_float(0x3F800000u | my);
    unsigned bad = 0, first = 0xFFFFFFFFu;
    for (unsigned i = 0; i < (1u << 20); ++i) {
        const float x = __uint_as_float(0x3F800000u | (x0 + i));
        const float q0 = quotient<0>(x, y);
        float q1;
        if (VARIANT == 0) q1 = quotient<1>(x, y);
        else if (VARIANT == 1) {   // 4 instructions: no refinement of the reciprocal
            const float r = __builtin_amdgcn_rcpf(y), q = x * r;
            q1 = __builtin_fmaf(__builtin_fmaf(-y, q, x), r, q);
        } else q1 = x * __builtin_amdgcn_rcpf(y);   // harness self-check: must mismatch often
        if (__float_as_uint(q0) != __float_as_uint(q1)) { ++bad; if (first == 0xFFFFFFFFu) first = x0 + i; }
    }
    if (bad) {
        atomicAdd(&counts[0], (unsigned long long)bad);
        const unsigned long long slot = atomicAdd(&counts[2], 1ull);
        if (slot < 8) counts[3 + slot] = ((unsigned long long)first << 32) | my;
    }
    if (threadIdx.x == 0) atomicAdd(&counts[1], 256ull << 20);
}
// v_rcp_f32 is exponent-invariant: rcp(m * 2^e) == rcp(m) * 2^-e for every significand and e in [-61, 61]
__global__ __launch_bounds__(256) void rcp_invariance_kernel(unsigned long long *counts) {
    const unsigned m = blockIdx.x * 256u + threadIdx.x;            // 2^23 threads
    const float y = __uint_as_float(0x3F800000u | m);
    const float r = __builtin_amdgcn_rcpf(y);
    unsigned bad = 0;
    for (int e = -61; e <= 61; ++e) bad += __float_as_uint(__builtin_amdgcn_rcpf(ldexpf(y, e))) != __float_as_uint(ldexpf(r, -e));
    if (bad) atomicAdd(&counts[0], (unsigned long long)bad);
}
hipError_t launch_divide_exhaustive(unsigned long long *counts, int slice, hipStream_t stream) {
    if (slice < 0) hipLaunchKernelGGL(rcp_invariance_kernel, dim3(1u << 15), dim3(256), 0, stream, counts);
    else if (slice >= 128) hipLaunchKernelGGL(divide_exhaustive_kernel<2>, dim3(4096), dim3(256), 0, stream, counts, (unsigned)slice - 128);
    else if (slice >= 64) hipLaunchKernelGGL(divide_exhaustive_kernel<1>, dim3(4096), dim3(256), 0, stream, counts, (unsigned)slice - 64);
    else hipLaunchKernelGGL(divide_exhaustive_kernel<0>, dim3(4096), dim3(256), 0, stream, counts, (unsigned)slice);
    return hipGetLastError();
}

}  // namespace nmf
