// nmf_kernels.hip -- dispatch between the fused kernel families and the small kernels of the update_div hot path
// (partial-slab reduction, normalisers, flat KL reduction, elementwise operators, padding) for gfx950.
//
// Reference semantics restated (not translated): cuda/nmf.cu:118-176 (half-steps),
// cuda/matrix.cu:127-250 (elementwise operators), cuda/matrix.cu:505-735 (reductions).
#include "nmf_device.h"

#include <atomic>
#include <cxxabi.h>
#include <string>

namespace nmf {
static std::atomic<int> g_record_kernels{0};
static thread_local std::string g_last_kernel;
void note_kernel(const void *host_fn, hipStream_t stream) {
    if (!g_record_kernels.load(std::memory_order_relaxed)) return;
    const char *m = hipKernelNameRefByPtr(host_fn, stream);
    if (!m) { g_last_kernel = "?"; return; }
    int st = 0;
    char *d = abi::__cxa_demangle(m, nullptr, nullptr, &st);
    g_last_kernel = (st == 0 && d) ? d : m;
    free(d);
}
}  // namespace nmf
extern "C" __attribute__((visibility("default"))) int nmf_debug_record_kernels(int on) { return nmf::g_record_kernels.exchange(on ? 1 : 0); }
extern "C" __attribute__((visibility("default"))) const char *nmf_debug_last_kernel(void) { return nmf::g_last_kernel.c_str(); }

namespace nmf {

// which kernel family serves a padded K: the 16-column kernel up to 512 (two workgroups per CU up to 256, one above; K <= 32 too since
// round 4: 98.5 against 91.1 TFLOP/s at 4096 x 65536 x 32), the 32-column v3 only when NMF_FUSED_VARIANT asks for it (K <= 256)
static bool use_pair(int Kp) { return Kp > kMaxK16; }   // two waves per 16 owned columns, K split between them (nmf_pair16.hip)
static bool use_k16(int Kp) { return !use_pair(Kp) && (Kp > 256 || fused_variant() == 0); }

__global__ __launch_bounds__(256) void zero_kernel(uint4 *__restrict__ p, size_t n16) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = uint4{0u, 0u, 0u, 0u};
}
hipError_t launch_zero(void *p, size_t bytes, hipStream_t stream) {
    if (!bytes) return hipSuccess;
    if ((bytes & 15) || ((uintptr_t)p & 15)) return hipMemsetAsync(p, 0, bytes, stream);
    hipLaunchKernelGGL(zero_kernel, dim3(ew_grid(bytes / 16)), dim3(256), 0, stream, (uint4 *)p, bytes / 16);
    return hipGetLastError();
}

hipError_t launch_fused_step(const FusedArgs &a, bool wstep, hipStream_t stream) {
    if ((a.Mp | a.Np | a.Kp) & 31) return hipErrorInvalidValue;
    if (a.nsplit < 1 || (a.nsplit > 1 && !a.partial)) return hipErrorInvalidValue;
    if (a.batch != 1 && !use_k16(a.Kp)) return hipErrorInvalidValue;   // blockIdx.y = pair exists on the 16-column kernel only
    if (use_pair(a.Kp)) return launch_fused_pair(a, wstep, stream);
    return use_k16(a.Kp) ? launch_fused16(a, wstep, stream) : launch_fused32(a, wstep, stream);
}

hipError_t launch_check(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc, double *part, hipStream_t stream,
                        int batch, size_t strideW, size_t strideH, int nsplit) {
    if ((Mp | Np | Kp) & 31) return hipErrorInvalidValue;
    if (batch < 1 || batch > 65535) return hipErrorInvalidValue;
    if (nsplit < 1 || (nsplit > 1 && !use_k16(Kp))) return hipErrorInvalidValue;
    if (use_pair(Kp)) {
        for (int b = 0; b < batch; ++b) {
            hipError_t e = launch_check_pair(W + (size_t)b * strideW, H + (size_t)b * strideH, X, Mp, Np, Kp, Kc, part + 3 * (size_t)check_num_groups(Np, Kp) * b, stream);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    return use_k16(Kp) ? launch_check16(W, H, X, Mp, Np, Kp, Kc, part, stream, batch, strideW, strideH, nsplit) : launch_check32(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH);
}

// two rows of H per wave and workgroup (one per half-wave): the Mp/64 workgroups of a split must cover all Kp rows
bool fused_streams_vsum(int Mp, int Kp) { return use_k16(Kp) && (size_t)((Mp + 63) / 64) * 8 >= (size_t)Kp; }
bool fused_takes_batch(int Kp) { return use_k16(Kp); }
int fused_cols_per_group(int Kp) { return use_pair(Kp) ? 32 : (use_k16(Kp) ? 64 : 128); }
// K in HBM: padded to 32 like the reference (PAD_MULT, cuda/matrix.cuh:7), nothing coarser up to 576 -- the 16-column kernel has an
// instantiation for every multiple of 16 up to 576 (fused16_compute_k).  The 32-column kernel (NMF_FUSED_VARIANT=3,
// an A/B switch) only exists for 32 / 64 / 128 / 256.  0 = not supported.
int fused_pad_k(int K) {
    const int k32 = pad32(K);
    if (k32 <= 32) return 32;
    if (k32 <= 256 && fused_variant() != 0) { int kt = k32 / 32, p = 1; while (p < kt) p <<= 1; return 32 * p; }
    if (k32 <= kMaxK16) return k32;
    return K <= kMaxFusedK ? pair_pad_k(K) : 0;   // the wave-pair kernel: 64 in HBM (whole staged pieces), 32 in its MFMAs (pair_compute_k)
}
int fused_compute_k(int K) {
    const int kp = fused_pad_k(K);
    if (kp && use_pair(kp)) return pair_compute_k(K);
    if (!kp || !use_k16(kp)) return kp;
    return fused16_compute_k(K);
}
int check_num_groups(int Np, int Kp) { return (Np + fused_cols_per_group(Kp) - 1) / fused_cols_per_group(Kp); }

// ------------------------------------------------------------------ partial reduce + apply
template <bool WSTEP>
__global__ __launch_bounds__(256) void apply_partials_kernel(float *__restrict__ U, const float *__restrict__ P, int nsplit,
                                                             const float *__restrict__ nrm, const float *__restrict__ vsum_part,
                                                             size_t count, int Mp, int Kp, size_t ustride, const int *__restrict__ active) {
    const size_t b = blockIdx.y;   // pair of a batched solver
    if (active != nullptr && active[b] == 0) return;
    U += b * ustride; P += b * (size_t)nsplit * count;
    if (nrm) nrm += b * (size_t)Kp;
    if (vsum_part) vsum_part += b * (size_t)nsplit * Kp;
    int k_cached = -1;
    float n_cached = 1.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float s = P[i];
        int sp0 = 1;
        for (; sp0 + 4 <= nsplit; sp0 += 4) {   // fixed order, four loads in flight
            const float a0 = P[(size_t)sp0 * count + i], a1 = P[(size_t)(sp0 + 1) * count + i], a2 = P[(size_t)(sp0 + 2) * count + i], a3 = P[(size_t)(sp0 + 3) * count + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; sp0 < nsplit; ++sp0) s += P[(size_t)sp0 * count + i];
        const int k = WSTEP ? (int)(i / (size_t)Mp) : (int)(i % (size_t)Kp);
        if (k != k_cached) {
            k_cached = k;
            if (vsum_part) {   // row sums of H delivered per split by the W-step kernel: fixed-order sum, then set_epsilon
                float n = vsum_part[k];
                for (int sp = 1; sp < nsplit; ++sp) n += vsum_part[(size_t)sp * Kp + k];
                n_cached = clamp_eps(n);
            } else n_cached = nrm[k];
        }
        U[i] = U[i] * (s / n_cached);
    }
}


hipError_t launch_apply_partials(float *U, const float *partials, int nsplit, const float *norm, int Mp, int Np, int Kp,
                                 bool wstep, hipStream_t stream, const float *vsum_part, int batch, size_t ustride, const int *active) {
    const size_t count = wstep ? (size_t)Mp * Kp : (size_t)Kp * Np;
    if ((!norm && !vsum_part) || (vsum_part && !wstep) || batch < 1 || batch > 65535) return hipErrorInvalidValue;
    const dim3 grid(ew_grid(count), (unsigned)batch);
    if (wstep) hipLaunchKernelGGL(apply_partials_kernel<true>, grid, dim3(256), 0, stream, U, partials, nsplit, norm, vsum_part, count, Mp, Kp, ustride, active);
    else       hipLaunchKernelGGL(apply_partials_kernel<false>, grid, dim3(256), 0, stream, U, partials, nsplit, norm, vsum_part, count, Mp, Kp, ustride, active);
    return hipGetLastError();
}

// Sum over a 1024-thread workgroup of per-thread partials (thread t holds rows t, t + 1024, ... of one column): wave shuffle
// trees, then the 16 wave results in order.  col_sums_kernel and apply_w_colsum_kernel both end in this, so the normaliser
// of an H-step has the same bits whether it was left by the previous W-step's apply or computed afresh after an upload
// (resuming from saved factors equals the uninterrupted run, tests/test_gpu_update_div.py::test_resume_...).
__device__ __forceinline__ float column_total_1024(float partial) {
    __shared__ float red[16];
    partial = wave_sum(partial);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = partial;
    __syncthreads();
    float tot = 0.f;
    for (int w = 0; w < 16; ++w) tot += red[w];
    return tot;
}

// One workgroup per column k of W (rows are contiguous): W[:,k] *= (sum_s P[s][:,k]) / n_k, and the column sum of the new
// values, reduced in a fixed order (per-thread strided partial -> wave shuffle tree -> 16 waves), clamped, goes to norm_out[k].
__global__ __launch_bounds__(1024) void apply_w_colsum_kernel(float *__restrict__ W, const float *__restrict__ P, int nsplit,
                                                              const float *__restrict__ nrm, const float *__restrict__ vsum_part,
                                                              int Mp, int Kp, float *__restrict__ norm_out, size_t wstride, const int *__restrict__ active) {
    const int k = blockIdx.x;
    const size_t b = blockIdx.y;   // pair of a batched solver
    if (active != nullptr && active[b] == 0) return;
    W += b * wstride; P += b * (size_t)nsplit * Mp * Kp; norm_out += b * (size_t)Kp;
    if (nrm) nrm += b * (size_t)Kp;
    if (vsum_part) vsum_part += b * (size_t)nsplit * Kp;
    float n;
    if (vsum_part) {
        n = vsum_part[k];
        for (int sp = 1; sp < nsplit; ++sp) n += vsum_part[(size_t)sp * Kp + k];
        n = clamp_eps(n);
    } else n = nrm[k];
    const size_t count = (size_t)Mp * Kp, col = (size_t)k * Mp;
    float acc = 0.f;
    // 32-256 workgroups have nothing to hide an L2 round trip behind but their own loads: four rows per thread at a time, the slabs of
    // all four fetched eight deep before the first add (worth 2-3 us per iteration: profiles/r04_apply_w_colsum_ab.log).  The sums keep
    // their fixed order: slabs 0, 1, 2, ... per element, rows t, t + 1024, ... per thread.
    constexpr int R = 4, S = 8;
    for (int i0 = threadIdx.x; i0 < Mp; i0 += R * 1024) {
        float s[R], wv[R];
        const float *__restrict__ p[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r * 1024;
            p[r] = P + col + (i < Mp ? i : i0);   // rows past the end re-read row i0 and are dropped below
            wv[r] = W[col + (i < Mp ? i : i0)];
            s[r] = 0.f;
        }
        int sp = 0;
        // slab 0 starts the sum (assigned, not added to zero: the bits of the one-slab-at-a-time loop this replaces)
#define NMF_APPLY_BATCH(B)                                                                        \
        for (; sp + (B) <= nsplit; sp += (B)) {                                                   \
            float a[R][B];                                                                        \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                         \
                _Pragma("unroll") for (int e = 0; e < (B); ++e) a[r][e] = p[r][(size_t)(sp + e) * count]; \
            _Pragma("unroll") for (int r = 0; r < R; ++r)                                         \
                _Pragma("unroll") for (int e = 0; e < (B); ++e) s[r] = (sp == 0 && e == 0) ? a[r][0] : s[r] + a[r][e]; \
        }
        NMF_APPLY_BATCH(S)
        NMF_APPLY_BATCH(4)
        NMF_APPLY_BATCH(1)
#undef NMF_APPLY_BATCH
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = i0 + r * 1024;
            if (i < Mp) {
                const float w = __fmul_rn(wv[r], s[r] / n);   // _rn: the product that is stored is the one that is summed (no fma contraction)
                W[col + i] = w;
                acc = __fadd_rn(acc, w);
            }
        }
    }
    const float tot = column_total_1024(acc);
    if (threadIdx.x == 0) norm_out[k] = clamp_eps(tot);
}
hipError_t launch_apply_w_colsum(float *W, const float *partials, int nsplit, const float *norm, const float *vsum_part, int Mp, int Kp,
                                 float *norm_out, hipStream_t stream, int batch, size_t wstride, const int *active) {
    if ((!norm && !vsum_part) || !norm_out || batch < 1 || batch > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(apply_w_colsum_kernel, dim3(Kp, (unsigned)batch), dim3(1024), 0, stream, W, partials, nsplit, norm, vsum_part, Mp, Kp, norm_out, wstride, active);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void sum_partials_kernel(float *__restrict__ out, const float *__restrict__ P, int nsplit, size_t count,
                                                           const float *__restrict__ vsum_part, int Kp, int Mp, int q_valid) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        if (q_valid > 0 && (int)(i % (size_t)Mp) >= q_valid) { out[i] = 0.f; continue; }
        float s = P[i];
        int sp = 1;
        for (; sp + 4 <= nsplit; sp += 4) {   // fixed order, four loads in flight
            const float a0 = P[(size_t)sp * count + i], a1 = P[(size_t)(sp + 1) * count + i], a2 = P[(size_t)(sp + 2) * count + i], a3 = P[(size_t)(sp + 3) * count + i];
            s += a0; s += a1; s += a2; s += a3;
        }
        for (; sp < nsplit; ++sp) s += P[(size_t)sp * count + i];
        out[i] = s;
    }
    if (vsum_part) {
        for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < (size_t)Kp; k += (size_t)gridDim.x * 256) {
            float n = vsum_part[k];
            for (int sp = 1; sp < nsplit; ++sp) n += vsum_part[(size_t)sp * Kp + k];
            out[count + k] = n;
        }
    }
}
hipError_t launch_sum_partials(float *psum, const float *partials, int nsplit, size_t count, hipStream_t stream, const float *vsum_part, int Kp, int Mp, int q_valid) {
    if (q_valid >= Mp) q_valid = 0;
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ew_grid(count)), dim3(256), 0, stream, psum, partials, nsplit, count, vsum_part, Kp, Mp > 0 ? Mp : 1, q_valid);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void apply_w_kernel(float *__restrict__ W, const float *__restrict__ psum, const float *__restrict__ hsum,
                                                      size_t count, int Mp) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i / (size_t)Mp);
        W[i] = W[i] * (psum[i] / clamp_eps(hsum[k]));
    }
}
hipError_t launch_apply_w(float *W, const float *psum, const float *hsum, int Mp, int Kp, hipStream_t stream) {
    const size_t count = (size_t)Mp * Kp;
    hipLaunchKernelGGL(apply_w_kernel, dim3(ew_grid(count)), dim3(256), 0, stream, W, psum, hsum, count, Mp);
    return hipGetLastError();
}

// out3[v] = sum_g part[3g + v], fixed order (one workgroup)
__global__ __launch_bounds__(256) void check_final_kernel(const double *__restrict__ part, int ngroups, double *__restrict__ out3) {
    double v0 = 0.0, v1 = 0.0, v2 = 0.0;
    for (int g = threadIdx.x; g < ngroups; g += 256) { v0 += part[3 * (size_t)g]; v1 += part[3 * (size_t)g + 1]; v2 += part[3 * (size_t)g + 2]; }
    block_reduce3(v0, v1, v2, out3, threadIdx.x);
}
hipError_t launch_check_final(const double *part, int ngroups, double *out3, hipStream_t stream) {
    hipLaunchKernelGGL(check_final_kernel, dim3(1), dim3(256), 0, stream, part, ngroups, out3);
    return hipGetLastError();
}

// ---- the X-only terms of the check, once per upload
__global__ __launch_bounds__(256) void x_consts_kernel(const float *__restrict__ x, size_t n, double *__restrict__ part) {
    double a = 0.0, b = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xv = x[i];
        if (xv > 0.f) { const double xd = (double)xv; a += xd * log(xd) - xd; b += xd; }   // padding is exactly 0; NaN fails the test like in the reference's loop
        else if (xv != xv) { a += (double)xv; b += (double)xv; }
    }
    block_reduce3(a, b, 0.0, part + 3 * (size_t)blockIdx.x, threadIdx.x);
}
hipError_t launch_x_consts(const float *X, size_t n, double *part, double *xc3, hipStream_t stream) {
    size_t g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > (size_t)kXConstGroups) g = kXConstGroups;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(x_consts_kernel, dim3((unsigned)g), dim3(256), 0, stream, X, n, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_check_final(part, (int)g, xc3, stream);
}

// ---- sum y = sum_k colsum(W)_k rowsum(H)_k in fp64, and the composition of the three terms
__global__ __launch_bounds__(256) void colsum64_kernel(const float *__restrict__ W, int Mp, double *__restrict__ out) {
    __shared__ double red[4];
    const float *__restrict__ a = W + (size_t)blockIdx.x * Mp;
    double s0 = 0.0, s1 = 0.0;
    int i = threadIdx.x;
    for (; i + 256 < Mp; i += 512) { s0 += (double)a[i]; s1 += (double)a[i + 256]; }
    if (i < Mp) s0 += (double)a[i];
    double s = wave_sum(s0 + s1);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// sum y = sum_n sum_k colsum(W)_k H[k, n]: one coalesced pass over H with the fp64 column sums of W as weights, fp64
// accumulation per thread, one partial per workgroup (fixed grid: reproducible)
__global__ __launch_bounds__(256) void ysum_kernel(const float *__restrict__ H, size_t n, int Kp, const double *__restrict__ wsum, double *__restrict__ ypart) {
    extern __shared__ double wl[];
    for (int k = threadIdx.x; k < Kp; k += 256) wl[k] = wsum[k];
    __syncthreads();
    double s0 = 0.0, s1 = 0.0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + stride < n; i += 2 * stride) {
        s0 += wl[i % (size_t)Kp] * (double)H[i];
        s1 += wl[(i + stride) % (size_t)Kp] * (double)H[i + stride];
    }
    if (i < n) s0 += wl[i % (size_t)Kp] * (double)H[i];
    __shared__ double tot[3];
    block_reduce3(s0 + s1, 0.0, 0.0, tot, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) ypart[blockIdx.x] = tot[0];
}
__global__ __launch_bounds__(256) void check_compose_kernel(const double *__restrict__ part, int ngroups, const double *__restrict__ ypart, int ny,
                                                            const double *__restrict__ xc3, double *__restrict__ out3) {
    double sy = 0.0, sxly = 0.0, sd = 0.0;
    for (int g = threadIdx.x; g < ny; g += 256) sy += ypart[g];
    for (int g = threadIdx.x; g < ngroups; g += 256) { sxly += part[3 * (size_t)g]; sd += part[3 * (size_t)g + 1]; }
    __shared__ double tot[3];
    block_reduce3(sy, sxly, sd, tot, threadIdx.x);
    __syncthreads();
    if (threadIdx.x == 0) {
        out3[0] = xc3[0] - 0.69314718055994530942 * tot[1] + tot[0];   // [sum x log x - x] - ln 2 [sum x log2 y] + [sum y]
        out3[1] = tot[2];
        out3[2] = xc3[1];
    }
}
hipError_t launch_check_compose(const double *part, int ngroups, const float *W, const float *H, int Mp, int Np, int Kp,
                                const double *xc3, double *scratch64, double *out3, hipStream_t stream) {
    double *wsum = scratch64, *ypart = scratch64 + Kp;
    const size_t n = (size_t)Kp * Np;
    size_t ny = (n + 256 * 32 - 1) / (256 * 32);
    if (ny > (size_t)kSum64Blocks) ny = kSum64Blocks;
    if (ny < 1) ny = 1;
    hipLaunchKernelGGL(colsum64_kernel, dim3(Kp), dim3(256), 0, stream, W, Mp, wsum);
    hipLaunchKernelGGL(ysum_kernel, dim3((unsigned)ny), dim3(256), (size_t)Kp * sizeof(double), stream, H, n, Kp, wsum, ypart);
    hipLaunchKernelGGL(check_compose_kernel, dim3(1), dim3(256), 0, stream, part, ngroups, ypart, (int)ny, xc3, out3);
    return hipGetLastError();
}

// generic flat version for the unfused path: x = X, y = WH (already clamped by the caller)
__global__ __launch_bounds__(256) void kl_reduce_kernel(const float *__restrict__ x, const float *__restrict__ y, size_t n, double *__restrict__ part) {
    double kl = 0.0, dabs = 0.0, xabs = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float xv = x[i], yv = y[i];
        if (xv > 0.f) {
            kl += (double)(xv * (logf(xv) - logf(yv)) - xv + yv);
            dabs += (double)fabsf(xv - yv);
            xabs += (double)fabsf(xv);
        }
    }
    block_reduce3(kl, dabs, xabs, part + 3 * (size_t)blockIdx.x, threadIdx.x);
}
int reduce_num_groups(size_t n) {
    size_t g = (n + 256 * 16 - 1) / (256 * 16);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}
hipError_t launch_kl_reduce(const float *x, const float *y, size_t n, double *part, hipStream_t stream) {
    hipLaunchKernelGGL(kl_reduce_kernel, dim3(reduce_num_groups(n)), dim3(256), 0, stream, x, y, n, part);
    return hipGetLastError();
}

// =====================================================================================
// Normalisers (sum_cols / sum_rows + set_epsilon, cuda/nmf.cu:134-135, 164-165)
// wave64 shuffle tree + LDS across the 4 waves; fixed summation order -> reproducible.
// =====================================================================================
__global__ __launch_bounds__(1024) void col_sums_kernel(const float *__restrict__ A, int rows, long ld, float *__restrict__ out, int clamp, size_t astride) {
    A += (size_t)blockIdx.y * astride; out += (size_t)blockIdx.y * gridDim.x;   // matrix b of a batch
    const float *__restrict__ a = A + (size_t)blockIdx.x * ld;
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 1024) s += a[i];
    const float tot = column_total_1024(s);
    if (threadIdx.x == 0) out[blockIdx.x] = clamp ? clamp_eps(tot) : tot;
}
hipError_t launch_col_sums(const float *A, int rows, int cols, long ld, float *out, bool clamp, hipStream_t stream, int batch, size_t astride) {
    if (batch < 1 || batch > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(col_sums_kernel, dim3(cols, (unsigned)batch), dim3(1024), 0, stream, A, rows, ld, out, clamp ? 1 : 0, astride);
    return hipGetLastError();
}

constexpr int kRowSumCols = 64;   // columns per workgroup in level 1 (>= 1024 workgroups at N = 65536)
int row_sum_blocks(int cols) { return (cols + kRowSumCols - 1) / kRowSumCols; }

// level 1: part[b*rows + k] = sum over this block's columns of A[k + col*ld]; fixed order per (b, k)
__global__ __launch_bounds__(256) void row_sums_l1_kernel(const float *__restrict__ A, int rows, int cols, long ld, float *__restrict__ part, size_t astride) {
    __shared__ float red[256];
    A += (size_t)blockIdx.y * astride; part += (size_t)blockIdx.y * gridDim.x * rows;   // matrix b of a batch
    const int c0 = blockIdx.x * kRowSumCols;
    const int c1 = (c0 + kRowSumCols < cols) ? c0 + kRowSumCols : cols;
    float *__restrict__ outp = part + (size_t)blockIdx.x * rows;
    if (rows >= 256) {
        for (int k = threadIdx.x; k < rows; k += 256) {
            const float *__restrict__ a = A + (size_t)k;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;   // four independent chains keep loads in flight
            int col = c0;
            for (; col + 4 <= c1; col += 4) {
                s0 += a[(size_t)(col + 0) * ld]; s1 += a[(size_t)(col + 1) * ld];
                s2 += a[(size_t)(col + 2) * ld]; s3 += a[(size_t)(col + 3) * ld];
            }
            for (; col < c1; ++col) s0 += a[(size_t)col * ld];
            outp[k] = (s0 + s1) + (s2 + s3);
        }
    } else {
        const int nsub = 256 / rows;             // column phases handled in parallel
        const int k = threadIdx.x % rows, sub = threadIdx.x / rows;
        float s = 0.f;
        if (sub < nsub)
            for (int col = c0 + sub; col < c1; col += nsub) s += A[(size_t)k + (size_t)col * ld];
        red[threadIdx.x] = s;
        __syncthreads();
        if (threadIdx.x < rows) {
            float tot = red[threadIdx.x];
            for (int q = 1; q < nsub; ++q) tot += red[threadIdx.x + q * rows];
            outp[threadIdx.x] = tot;
        }
    }
}
// level 2: out[k] = sum_b part[b*rows + k].  One workgroup per 32 rows; 32 groups of 32 lanes stride
// over the partial blocks, then a fixed-order LDS combine.
__global__ __launch_bounds__(1024) void row_sums_l2_kernel(const float *__restrict__ part, int rows, int nblk, float *__restrict__ out, int clamp) {
    __shared__ float red[32][33];
    part += (size_t)blockIdx.y * nblk * rows; out += (size_t)blockIdx.y * rows;   // matrix b of a batch
    const int kl = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kl;
    float s0 = 0.f, s1 = 0.f;
    if (k < rows) {
        int b = g;
        for (; b + 32 < nblk; b += 64) { s0 += part[(size_t)b * rows + k]; s1 += part[(size_t)(b + 32) * rows + k]; }
        if (b < nblk) s0 += part[(size_t)b * rows + k];
    }
    red[g][kl] = s0 + s1;
    __syncthreads();
    if (g == 0 && k < rows) {
        float tot = 0.f;
        for (int q = 0; q < 32; ++q) tot += red[q][kl];
        out[k] = clamp ? clamp_eps(tot) : tot;
    }
}
hipError_t launch_row_sums(const float *A, int rows, int cols, long ld, float *part, float *out, bool clamp, hipStream_t stream, int batch, size_t astride) {
    if (batch < 1 || batch > 65535) return hipErrorInvalidValue;
    const int nblk = row_sum_blocks(cols);
    hipLaunchKernelGGL(row_sums_l1_kernel, dim3(nblk, (unsigned)batch), dim3(256), 0, stream, A, rows, cols, ld, part, astride);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_sums_l2_kernel, dim3((rows + 31) / 32, (unsigned)batch), dim3(1024), 0, stream, part, rows, nblk, out, clamp ? 1 : 0);
    return hipGetLastError();
}

// =====================================================================================
// Elementwise operators (cuda/matrix.cu:127-250)
// =====================================================================================
__global__ __launch_bounds__(256) void set_epsilon_kernel(float *__restrict__ a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float v = a[i];
        if (v < kEps) a[i] = kEps;
    }
}
hipError_t launch_set_epsilon(float *a, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(set_epsilon_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, a, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vec_div_kernel(const float *a, const float *b, float *c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i] / b[i];
}
hipError_t launch_vec_div(const float *a, const float *b, float *c, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vec_div_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, a, b, c, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void vec_mul_kernel(const float *a, const float *b, float *c, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c[i] = a[i] * b[i];
}
hipError_t launch_vec_mul(const float *a, const float *b, float *c, size_t n, hipStream_t stream) {
    hipLaunchKernelGGL(vec_mul_kernel, dim3(ew_grid(n)), dim3(256), 0, stream, a, b, c, n);
    return hipGetLastError();
}

// c[i + j*ld] = a[i + j*ld] / b[BY_ROW ? i : j]
template <bool BY_ROW>
__global__ __launch_bounds__(256) void bcast_div_kernel(const float *a, const float *b, float *c, int rows, int cols, long ld) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % (size_t)rows), j = (int)(e / (size_t)rows);
        const size_t ix = (size_t)i + (size_t)j * ld;
        c[ix] = a[ix] / b[BY_ROW ? i : j];
    }
}
hipError_t launch_col_div(const float *a, const float *b, float *c, int rows, int cols, long ld, hipStream_t stream) {
    hipLaunchKernelGGL(bcast_div_kernel<true>, dim3(ew_grid((size_t)rows * cols)), dim3(256), 0, stream, a, b, c, rows, cols, ld);
    return hipGetLastError();
}
hipError_t launch_row_div(const float *a, const float *b, float *c, int rows, int cols, long ld, hipStream_t stream) {
    hipLaunchKernelGGL(bcast_div_kernel<false>, dim3(ew_grid((size_t)rows * cols)), dim3(256), 0, stream, a, b, c, rows, cols, ld);
    return hipGetLastError();
}

// =====================================================================================
// Padding
// =====================================================================================
__global__ __launch_bounds__(256) void pad_copy_kernel(float *__restrict__ dst, int rows_p, int cols_p, const float *__restrict__ src, int rows, int cols,
                                                       int clamp, unsigned *__restrict__ range_flag) {
    const size_t n = (size_t)rows_p * cols_p;
    bool out_of_range = false;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % (size_t)rows_p), j = (int)(e / (size_t)rows_p);
        float v = 0.f;
        if (i < rows && j < cols) {
            v = src[(size_t)i + (size_t)j * rows];
            if (clamp) v = clamp_eps(v);
            out_of_range |= !(v <= kDivSafeMax);      // NaN counts as out of range
        }
        dst[e] = v;
    }
    if (range_flag && out_of_range) atomicOr(range_flag, 1u);
}
hipError_t launch_pad_copy(float *dst, int rows_p, int cols_p, const float *src, int rows, int cols, bool clamp, unsigned *range_flag,
                           hipStream_t stream) {
    hipLaunchKernelGGL(pad_copy_kernel, dim3(ew_grid((size_t)rows_p * cols_p)), dim3(256), 0, stream, dst, rows_p, cols_p, src, rows, cols,
                       clamp ? 1 : 0, range_flag);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void unpad_copy_kernel(float *__restrict__ dst, int rows, int cols, const float *__restrict__ src, int rows_p) {
    const size_t n = (size_t)rows * cols;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        const int i = (int)(e % (size_t)rows), j = (int)(e / (size_t)rows);
        dst[e] = src[(size_t)i + (size_t)j * rows_p];
    }
}
hipError_t launch_unpad_copy(float *dst, int rows, int cols, const float *src, int rows_p, hipStream_t stream) {
    hipLaunchKernelGGL(unpad_copy_kernel, dim3(ew_grid((size_t)rows * cols)), dim3(256), 0, stream, dst, rows, cols, src, rows_p);
    return hipGetLastError();
}

}  // namespace nmf

// ---------------------------------------------------------------- the emulated communicator group's reduction (nmf_comm.h)
#include "nmf_comm.h"
namespace {
struct EmuPtrs { const void *p[NMF_EMU_MAX_RANKS]; };
template <typename T>
__global__ __launch_bounds__(256) void emu_sum_kernel(EmuPtrs src, int n, T *__restrict__ dst, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        T s = reinterpret_cast<const T *>(src.p[0])[i];
        for (int r = 1; r < n; ++r) s += reinterpret_cast<const T *>(src.p[r])[i];   // rank order: the same bits on every rank
        dst[i] = s;
    }
}
}  // namespace
hipError_t nmf_emu_sum_launch(const void *const *src, int n, void *dst, size_t count, bool f64, hipStream_t stream) {
    if (n < 1 || n > NMF_EMU_MAX_RANKS) return hipErrorInvalidValue;
    EmuPtrs ptrs = {};
    for (int r = 0; r < n; ++r) ptrs.p[r] = src[r];
    size_t grid = (count + 255) / 256;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    if (f64) hipLaunchKernelGGL(emu_sum_kernel<double>, dim3((unsigned)grid), dim3(256), 0, stream, ptrs, n, (double *)dst, count);
    else     hipLaunchKernelGGL(emu_sum_kernel<float>, dim3((unsigned)grid), dim3(256), 0, stream, ptrs, n, (float *)dst, count);
    return hipGetLastError();
}

// the completion ticket of nmf_comm_wait: everything enqueued on the stream before this has finished when the host sees `value`
namespace {
__global__ void flag_store_kernel(unsigned *word, unsigned value) {
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
}  // namespace
hipError_t nmf_flag_store_launch(unsigned *word_host_mapped, unsigned value, hipStream_t stream) {
    hipLaunchKernelGGL(flag_store_kernel, dim3(1), dim3(1), 0, stream, word_host_mapped, value);
    return hipGetLastError();
}
