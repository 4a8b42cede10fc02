// nmf_split16.hip -- dispatch of the split half-step (nmf_split16_impl.h) over its instantiations (nmf_split16_inst.hip), and the
// kernel that finishes a half-step whose reduction dimension was split over workgroups.
#include "nmf_split16_impl.h"

namespace nmf {

#define NMF_S16_EXTERN(KT, NW, OCC, DB) extern template hipError_t launch_split_k16<KT, NW, OCC, DB>(const SplitArgs &, bool, hipStream_t);
NMF_S16_ALL(NMF_S16_EXTERN)
#undef NMF_S16_EXTERN

// U[b][k,q] *= (sum_s slab[b][s][k,q]) / max(sum_s vpart[b][s][k], EPS)   (col_div/row_div + vec_mul, cuda/matrix.cu:174-250,
// behind sum_cols/sum_rows + set_epsilon): four consecutive elements per thread.  The slabs are summed in a fixed order,
// four at a time so that four loads are in flight per thread (a few dozen workgroups have no other way to hide an L2 miss).
__device__ __forceinline__ f32x4 slab_sum4(const float *__restrict__ p, size_t stride, int nsplit) {
    f32x4 s = *reinterpret_cast<const f32x4 *>(p);
    int sp = 1;
    for (; sp + 4 <= nsplit; sp += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4 *>(p + (size_t)sp * stride), a1 = *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 1) * stride);
        const f32x4 a2 = *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 2) * stride), a3 = *reinterpret_cast<const f32x4 *>(p + (size_t)(sp + 3) * stride);
        s += a0; s += a1; s += a2; s += a3;
    }
    for (; sp < nsplit; ++sp) s += *reinterpret_cast<const f32x4 *>(p + (size_t)sp * stride);
    return s;
}
template <bool WSTEP>
__global__ __launch_bounds__(256) void split_apply_kernel(float *__restrict__ U, const float *__restrict__ P, const float *__restrict__ vpart, int nsplit,
                                                          size_t count, size_t todo, size_t ustride, int Mp, int Kp, int q_valid, const int *__restrict__ active) {
    const int b = blockIdx.y;
    if (active != nullptr && active[b] == 0) return;
    float *__restrict__ Ub = U + (size_t)b * ustride;
    const float *__restrict__ Pb = P + (size_t)b * nsplit * count;
    const float *__restrict__ vb = vpart + (size_t)b * nsplit * Kp;
    for (size_t i = 4 * ((size_t)blockIdx.x * 256 + threadIdx.x); i < todo; i += 4 * (size_t)gridDim.x * 256) {
        if (WSTEP && (int)(i % (size_t)Mp) >= q_valid) continue;   // zero padding: no workgroup wrote these slab entries
        const f32x4 s = slab_sum4(Pb + i, count, nsplit);
        f32x4 u = *reinterpret_cast<const f32x4 *>(Ub + i);
        if (WSTEP) {   // W (Mp x Kp): the four elements share column k (Mp % 4 == 0)
            const int k = (int)(i / (size_t)Mp);
            float n = vb[k];              // same fixed order as slab_sum4, four loads in flight
            int sp = 1;
            for (; sp + 4 <= nsplit; sp += 4) {
                const float a0 = vb[(size_t)sp * Kp + k], a1 = vb[(size_t)(sp + 1) * Kp + k], a2 = vb[(size_t)(sp + 2) * Kp + k], a3 = vb[(size_t)(sp + 3) * Kp + k];
                n += a0; n += a1; n += a2; n += a3;
            }
            for (; sp < nsplit; ++sp) n += vb[(size_t)sp * Kp + k];
            n = clamp_eps(n);
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = u[e] * (s[e] / n);
        } else {       // H (Kp x Np): rows k .. k + 3 of one column
            const int k = (int)(i % (size_t)Kp);
            const f32x4 n = slab_sum4(vb + k, (size_t)Kp, nsplit);
#pragma unroll
            for (int e = 0; e < 4; ++e) u[e] = u[e] * (s[e] / clamp_eps(n[e]));
        }
        *reinterpret_cast<f32x4 *>(Ub + i) = u;
    }
}

hipError_t launch_split_apply(float *U, const float *partials, const float *vpart, int nsplit, int Mp, int Np, int Kp, int q_valid, bool wstep,
                              int batch, size_t ustride, const int *active, hipStream_t stream) {
    const size_t count = wstep ? (size_t)Mp * Kp : (size_t)Kp * Np;       // slab stride
    const size_t todo = wstep ? count : (size_t)Kp * q_valid;             // H: the valid columns are a prefix
    size_t g = (todo / 4 + 255) / 256;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    const dim3 grid((unsigned)g, (unsigned)batch);
    if (wstep) hipLaunchKernelGGL(split_apply_kernel<true>, grid, dim3(256), 0, stream, U, partials, vpart, nsplit, count, todo, ustride, Mp, Kp, q_valid, active);
    else       hipLaunchKernelGGL(split_apply_kernel<false>, grid, dim3(256), 0, stream, U, partials, vpart, nsplit, count, todo, ustride, Mp, Kp, q_valid, active);
    return hipGetLastError();
}

size_t split_step_lds_bytes(int Kp, int nw, bool db) { return ((size_t)(db ? 2 : 1) * nw * Kp * kLdv + nw * kXs16Floats) * sizeof(float); }
bool split_step_supports(int Kp) { return Kp >= 32 && Kp <= 256 && (Kp % 32) == 0; }
// the K the split kernel computes on for a logical K <= 256: the next multiple of 16, at least 32
int split_compute_k(int K) { return K <= 32 ? 32 : (K <= 256 ? ((K + 15) & ~15) : 0); }

hipError_t launch_split_step(const SplitArgs &a, bool wstep, hipStream_t stream) {
    const int nw = wstep ? a.nw_w : a.nw_h;
    const int kc = a.Kc > 0 ? a.Kc : a.Kp;
    if ((kc % 16) || kc < 32 || kc > 256 || a.Kp != ((kc + 31) & ~31)) return hipErrorInvalidValue;
    if ((nw != 4 && nw != 8) || (nw == 8 && kc != 64)) return hipErrorInvalidValue;
    if ((a.Mv & 31) || (a.Nv & 31) || a.Mv > a.Mp || a.Nv > a.Np || a.Mv <= 0 || a.Nv <= 0) return hipErrorInvalidValue;
    if (((a.Mp | a.Np) & 127) || ((wstep ? a.Np : a.Mp) % (32 * nw))) return hipErrorInvalidValue;   // whole superchunks: the solver pads
    const bool partial = a.nsplit > 1 || a.force_partial;
    if (a.nsplit < 1 || a.batch < 1 || (partial && (!a.partials || !a.vpart)) || (!partial && !a.U_out)) return hipErrorInvalidValue;
    if ((size_t)a.Mp * (size_t)a.Kp >= ((size_t)1 << 30) || (size_t)a.Kp * (size_t)a.Np >= ((size_t)1 << 30)) return hipErrorInvalidValue;   // 32-bit lane offsets
    const int kt = kc / 16;
    if (kt <= 4) {
        if (nw == 8) return launch_split_k16<4, 8, 2, true>(a, wstep, stream);
        switch (kt) {
            case 2: return launch_split_k16<2, 4, 2, true>(a, wstep, stream);
            case 3: return launch_split_k16<3, 4, 2, true>(a, wstep, stream);
            default: return launch_split_k16<4, 4, 2, true>(a, wstep, stream);
        }
    }
    if (kt <= 8) {
        switch (kt) {
            case 5: return a.single_image ? launch_split_k16<5, 4, 2, false>(a, wstep, stream) : launch_split_k16<5, 4, 1, true>(a, wstep, stream);
            case 6: return a.single_image ? launch_split_k16<6, 4, 2, false>(a, wstep, stream) : launch_split_k16<6, 4, 1, true>(a, wstep, stream);
            case 7: return a.single_image ? launch_split_k16<7, 4, 2, false>(a, wstep, stream) : launch_split_k16<7, 4, 1, true>(a, wstep, stream);
            default: return a.single_image ? launch_split_k16<8, 4, 2, false>(a, wstep, stream) : launch_split_k16<8, 4, 1, true>(a, wstep, stream);
        }
    }
    switch (kt) {
        case 9: return launch_split_k16<9, 4, 1, false>(a, wstep, stream);
        case 10: return launch_split_k16<10, 4, 1, false>(a, wstep, stream);
        case 11: return launch_split_k16<11, 4, 1, false>(a, wstep, stream);
        case 12: return launch_split_k16<12, 4, 1, false>(a, wstep, stream);
        case 13: return launch_split_k16<13, 4, 1, false>(a, wstep, stream);
        case 14: return launch_split_k16<14, 4, 1, false>(a, wstep, stream);
        case 15: return launch_split_k16<15, 4, 1, false>(a, wstep, stream);
        case 16: return launch_split_k16<16, 4, 1, false>(a, wstep, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nmf
