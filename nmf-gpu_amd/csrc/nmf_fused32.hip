// nmf_fused32.hip -- the 32-column fused half-step family (v_mfma_f32_32x32x2_f32): v3, the kernel for K <= 32 and
// NMF_FUSED_VARIANT=3; v1, the first chunk-serial kernel, kept as the 64-bit-addressing fallback and for the
// ablation probes; their KL check.  The production kernel for K >= 64 is in nmf_fused16_impl.h.
//
// Written for wave64 + the exact-fp32 MFMA v_mfma_f32_32x32x2_f32.  Operand / result maps
// used everywhere below (lane l: c = l & 31, h = l >> 5):
//     A operand : one float = A[row c][k = h]
//     B operand : one float = B[k = h][col c]
//     C/D tile  : reg r (0..15) = D[row rho(r) + 4h][col c],  rho(r) = (r & 3) + 8 (r >> 2)
// Consequence exploited by the fused kernels: register r of a finished 32x32 tile IS a valid B
// operand of a following MFMA whose two k indices are rows rho(r) and rho(r)+4 of that tile, so
// the quotient Z = X ./ max(W*H, EPS) feeds the second GEMM of a half-step straight from the
// accumulator registers and never exists in LDS or HBM.
//
// Reference semantics restated (not translated): cuda/nmf.cu:118-176 (half-steps),
// cuda/matrix.cu:97-250 (operators), cuda/matrix.cu:505-735 (reductions).
#include "nmf_device.h"

namespace nmf {

// =====================================================================================
// Fused half-step, first generation (v1)
// =====================================================================================
template <int KT, bool WSTEP>
__device__ __forceinline__ void stage_load(f32x4 (&st)[KT], const float *__restrict__ V, long ldv, int p0, int tid) {
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int f = tid + q * 256;
        if (!WSTEP) {   // V = W (p contiguous): K rows of 32 floats
            const int k = f >> 3, i4 = f & 7;
            st[q] = *reinterpret_cast<const f32x4 *>(V + (size_t)(p0 + 4 * i4) + (size_t)k * ldv);
        } else {        // V = H (k contiguous): 32 columns of K floats.  8 lanes cover 128 B of one column,
                        // the next 8 lanes the next column: full lines from HBM and, with the 33-float LDS
                        // rows, the transposing ds_write_b32 below hit 32 distinct banks.
            const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31;
            st[q] = *reinterpret_cast<const f32x4 *>(V + (size_t)(4 * k4) + (size_t)(p0 + i) * ldv);
        }
    }
}

template <int KT, bool WSTEP>
__device__ __forceinline__ void stage_store(const f32x4 (&st)[KT], float *__restrict__ vl, int tid) {
#pragma unroll
    for (int q = 0; q < KT; ++q) {
        const int f = tid + q * 256;
        if (!WSTEP) {
            const int k = f >> 3, i4 = f & 7;
#pragma unroll
            for (int c = 0; c < 4; ++c) vl[k * kLdv + 4 * i4 + c] = st[q][c];
        } else {
            const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31;
#pragma unroll
            for (int c = 0; c < 4; ++c) vl[(4 * k4 + c) * kLdv + i] = st[q][c];
        }
    }
}

// X tile of chunk p0 in the accumulator layout: xr[r] = X(p0 + rho(r) + 4h, q0 + c)
template <bool WSTEP>
__device__ __forceinline__ void load_x(float (&xr)[16], const float *__restrict__ X, long ldx, int p0, int q0, int c, int h) {
    if (!WSTEP) {   // X(p,q) = X[p + q*ld]: 4 consecutive p per lane per group
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(X + (size_t)(p0 + 8 * g + 4 * h) + (size_t)(q0 + c) * ldx);
            xr[4 * g + 0] = v[0]; xr[4 * g + 1] = v[1]; xr[4 * g + 2] = v[2]; xr[4 * g + 3] = v[3];
        }
    } else {        // X(p,q) = X[q + p*ld]: lanes run along q, coalesced
#pragma unroll
        for (int r = 0; r < 16; ++r) xr[r] = X[(size_t)(q0 + c) + (size_t)(p0 + rho(r) + 4 * h) * ldx];
    }
}

// B operands of product 1, resident for the whole kernel: ub[s] = U(k = 2s + h, q0 + c)
template <int KT, bool WSTEP>
__device__ __forceinline__ void load_u(float (&ub)[KT * 16], const float *__restrict__ U, long ldu, int q0, int c, int h) {
#pragma unroll
    for (int s = 0; s < KT * 16; ++s) {
        const int k = 2 * s + h;
        ub[s] = WSTEP ? U[(size_t)(q0 + c) + (size_t)k * ldu] : U[(size_t)k + (size_t)(q0 + c) * ldu];
    }
}

// Product 1: S(32 p x 32 q) = V_chunk * U_slice, one dependent chain of KT*16 MFMAs (the 32x32x2 f32
// MFMA has issue interval = dependent latency = 64 cycles, so a single chain runs at full rate as
// long as its A operand is already in a register).  The A operands come from LDS through a ring of
// kRing registers loaded kRing MFMAs (>= 512 cycles) ahead of their use; hipcc otherwise emits
// ds_read -> s_waitcnt lgkmcnt(0) -> MFMA and exposes the LDS latency on every pair.
template <int KT>
__device__ __forceinline__ f32x16 product1(const float (&ub)[KT * 16], const float *__restrict__ vb, int c, int h) {
    constexpr int N = KT * 16;
    constexpr int D = (N < kRing) ? N : kRing;
    const float *__restrict__ base = vb + h * kLdv + c;   // operand of step ss: base[2*ss*kLdv]
    float a[D];
#pragma unroll
    for (int i = 0; i < D; ++i) a[i] = base[2 * i * kLdv];
    f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < N; ++ss) {
        s = NMF_MFMA(a[ss % D], ub[ss], s);
        if (ss + D < N) a[ss % D] = base[2 * (ss + D) * kLdv];
        //__builtin_amdgcn_sched_group_barrier(NMF_SG_MFMA, 1, 0);
        //__builtin_amdgcn_sched_group_barrier(NMF_SG_DS_READ, 1, 0);
    }
    return s;
}

template <int KT>
__device__ __forceinline__ f32x16 product1_nolds(const float (&ub)[KT * 16], float av) {
    f32x16 s = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ss = 0; ss < KT * 16; ++ss) s = NMF_MFMA(av, ub[ss], s);
    return s;
}

// Epilogue shared by both kernel versions: lane holds Acc(k = 32t + rho(r) + 4h, q0 + c).
template <int KT, bool WSTEP, bool PARTIAL>
__device__ __forceinline__ void fused_epilogue(const FusedArgs &a, const f32x16 (&acc)[KT], int split, int q0, int c, int h, long ldu) {
    if (PARTIAL) {
        const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
        float *__restrict__ out = a.partials + (size_t)split * slab;
        if (!WSTEP) {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    f32x4 v = {acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
                    *reinterpret_cast<f32x4 *>(out + (size_t)(32 * t + 8 * g + 4 * h) + (size_t)(q0 + c) * ldu) = v;
                }
        } else {
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    out[(size_t)(q0 + c) + (size_t)(32 * t + rho(r) + 4 * h) * ldu] = acc[t][r];
        }
    } else {
        float *__restrict__ Uo = a.U_out;
        const float *__restrict__ nrm = a.norm;
        if (!WSTEP) {   // H[k,n] = H[k,n] * (WtZ[k,n] / sumW[k])   (col_div then vec_mul, cuda/nmf.cu:142-145)
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int k = 32 * t + 8 * g + 4 * h;
                    float *p = Uo + (size_t)k + (size_t)(q0 + c) * ldu;
                    f32x4 u = *reinterpret_cast<const f32x4 *>(p);
                    const f32x4 n4 = *reinterpret_cast<const f32x4 *>(nrm + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) u[e] = u[e] * (acc[t][4 * g + e] / n4[e]);
                    *reinterpret_cast<f32x4 *>(p) = u;
                }
        } else {        // W[m,k] = W[m,k] * (ZHt[m,k] / sumH[k])   (row_div then vec_mul, cuda/nmf.cu:172-175)
#pragma unroll
            for (int t = 0; t < KT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = 32 * t + rho(r) + 4 * h;
                    float *p = Uo + (size_t)(q0 + c) + (size_t)k * ldu;
                    *p = *p * (acc[t][r] / nrm[k]);
                }
        }
    }
}

// v1: one chunk at a time (product 1, then product 2), two LDS buffers.  Kept for A/B timing
// (NMF_FUSED_VARIANT=1); the production kernel is fused_step_kernel below.
// ABL (ablation bitmask, timing probes only; results are garbage when non-zero):
//   1 = no divide, 2 = MFMA A operands not read from LDS, 4 = no staging / barrier / X loads,
//   8 = no barrier only, 16 = no X loads only, 32 = no V staging (global load + LDS write) only
template <int KT, bool WSTEP, bool PARTIAL, int ABL = 0>
__global__ __launch_bounds__(256, 1) void fused_step_kernel_v1(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KP = KT * 32;
    constexpr int VBUF = KP * kLdv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int P = WSTEP ? a.Np : a.Mp;   // streamed / reduced dimension
    const int Q = WSTEP ? a.Mp : a.Np;   // owned dimension
    const int nsplit = a.nsplit;
    const int split = blockIdx.x % nsplit;   // workgroups of one split share the V stream (same XCD under round-robin)
    const int qblk = blockIdx.x / nsplit;
    int q0 = (qblk * 4 + wave) * 32;
    const bool active = q0 < Q;
    if (!active) q0 = Q - 32;                // tail wave: recompute a valid slice, store nothing
    const float *__restrict__ V = WSTEP ? a.H : a.W;
    const float *__restrict__ U = WSTEP ? a.W : a.H;
    const long ldv = WSTEP ? a.Kp : a.Mp, ldu = WSTEP ? a.Mp : a.Kp, ldx = a.Mp;
    const int nchunks = P / 32;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * cps;
    const int c_end = (c_begin + cps < nchunks) ? (c_begin + cps) : nchunks;

    float ub[KT * 16];
    load_u<KT, WSTEP>(ub, U, ldu, q0, c, h);

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (c_begin < c_end) {
        f32x4 st[KT];
        float xr[16];
        stage_load<KT, WSTEP>(st, V, ldv, c_begin * 32, tid);
        load_x<WSTEP>(xr, a.X, ldx, c_begin * 32, q0, c, h);
        stage_store<KT, WSTEP>(st, smem, tid);
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int par = (ch - c_begin) & 1;
            const float *__restrict__ vb = smem + par * VBUF;
            float *__restrict__ vn = smem + (par ^ 1) * VBUF;
            const bool more = ch + 1 < c_end;
            if (more && !(ABL & (4 | 32))) stage_load<KT, WSTEP>(st, V, ldv, (ch + 1) * 32, tid);

            // product 1: S(32 p x 32 q) = V_chunk * U_slice, reduction over K
            const f32x16 s = (ABL & 2) ? product1_nolds<KT>(ub, xr[0]) : product1<KT>(ub, vb, c, h);

            // product 2: Acc(K x 32 q) += V_chunk' * Z with Z = X ./ max(S, EPS) (set_epsilon + vec_div,
            // cuda/nmf.cu:128-131) taken straight from the accumulator layout: z(r) is the B operand for
            // the k-pair (rho(r), rho(r)+4).  The IEEE divide of row r+1, the LDS writes of the next
            // chunk and the operand prefetch are interleaved with the KT independent MFMAs of row r.
            constexpr int E = 16 * KT;
            constexpr int D2 = (E < kRing) ? E : kRing;
            const float *__restrict__ base2 = vb + c * kLdv + 4 * h;   // operand (r,t): base2[32*t*kLdv + rho(r)]
            float a2[D2];
#pragma unroll
            for (int e = 0; e < D2; ++e) a2[e] = (ABL & 2) ? xr[e % 16] : base2[32 * (e % KT) * kLdv + rho(e / KT)];
            float zc = (ABL & 1) ? xr[0] + s[0] : xr[0] / clamp_eps(s[0]);
            if (ABL & 64) {   // probe: product 2 as KT chains of 16 dependent MFMAs instead of round-robin
                float zz[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) zz[r] = xr[r] + s[r];
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t] = NMF_MFMA(xr[(r + t) % 16], zz[r], acc[t]);
            } else
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float zn = 0.f;
                if (r + 1 < 16) zn = (ABL & 1) ? xr[r + 1] + s[r + 1] : xr[r + 1] / clamp_eps(s[r + 1]);
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const int e = r * KT + t;
                    acc[t] = NMF_MFMA(a2[e % D2], zc, acc[t]);
                    if (e + D2 < E && !(ABL & 2)) a2[e % D2] = base2[32 * ((e + D2) % KT) * kLdv + rho((e + D2) / KT)];
                    //__builtin_amdgcn_sched_group_barrier(NMF_SG_MFMA, 1, 0);
                    //__builtin_amdgcn_sched_group_barrier(NMF_SG_DS_READ, 1, 0);
                    //__builtin_amdgcn_sched_group_barrier(NMF_SG_VALU, 3, 0);
                }
                zc = zn;
            }
            if (more && !(ABL & (4 | 16))) load_x<WSTEP>(xr, a.X, ldx, (ch + 1) * 32, q0, c, h);
            if (more && !(ABL & (4 | 32))) stage_store<KT, WSTEP>(st, vn, tid);
            if (!(ABL & (4 | 8))) __syncthreads();
        }
    }
    if (!active) return;

    fused_epilogue<KT, WSTEP, PARTIAL>(a, acc, split, q0, c, h, ldu);
}

// =====================================================================================
// v3: the production fused half-step.  Same chunk-serial structure as v1, rebuilt around what the
// micro-probes (launch_mfma_valu_probe) measured for a lone wave per SIMD next to f32 MFMAs:
//   * VALU and VMEM issue is NOT hidden (+4..5 cycles per VALU, +~13 for the first of a group,
//     +~20 per coalesced global load, hundreds for a lane-strided one),
//   * ds_read / ds_write / SALU issue IS hidden.
// Hence: no per-read LDS address arithmetic (one base VGPR + 16-bit immediate offsets, reads kept
// single by `volatile`), all divides of a chunk in one VALU block, the X tile fetched with four
// fully coalesced 16-B loads and re-laid into the accumulator layout through a private LDS patch,
// global addresses as uniform base + 32-bit lane offset.
// =====================================================================================

// v3 assigns the two k indices of product-1 step s to k = s (lanes 0-31) and k = s + 16*KT (lanes 32-63), so that a lane's
// B operands are contiguous in k: the H-step loads them as 16-B pieces (4x fewer lane-strided loads than k = 2s + h).
template <int KT, bool WSTEP>
__device__ __forceinline__ void load_u_split(float (&ub)[KT * 16], const float *__restrict__ U, long ldu, int q0, int c, int h) {
    constexpr int N1 = KT * 16;
    if (!WSTEP) {
        const float *__restrict__ col = U + (size_t)(N1 * h) + (size_t)(q0 + c) * ldu;
#pragma unroll
        for (int s4 = 0; s4 < N1 / 4; ++s4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 4 * s4);
            ub[4 * s4] = v[0]; ub[4 * s4 + 1] = v[1]; ub[4 * s4 + 2] = v[2]; ub[4 * s4 + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int s = 0; s < N1; ++s) ub[s] = U[(size_t)(q0 + c) + (size_t)(s + N1 * h) * ldu];
    }
}

// STAMP = true: diagnostic build only (never the shipped path): s_memtime stamps around the five segments of a
// chunk, summed per wave and written to a.partials as 5 x uint64 per wave; the results of the step stay valid.
#define NMF_STAMP(var)                                                                                     \
    do {                                                                                                   \
        if (STAMP) {                                                                                       \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");                        \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
    } while (0)
// CHECK = true: the KL / rel-L1 convergence check (product 1 only; reduce1d_div / reduce1d_diff, cuda/matrix.cu:505-640):
// one triple {KL, sum|x-y|, sum|x|} per workgroup into chk_part.
template <int KT, bool WSTEP, bool PARTIAL, int DIV, bool STAMP = false, bool CHECK = false>
__global__ __launch_bounds__(256, 1) void fused_step_kernel_v3(FusedArgs a, double *__restrict__ chk_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VBUF = KT * 32 * kLdv;
    constexpr int N1 = KT * 16;
    constexpr int D = (N1 < kRing) ? N1 : kRing;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, c = lane & 31, h = lane >> 5;
    const int P = WSTEP ? a.Np : a.Mp;   // streamed / reduced dimension
    const int Q = WSTEP ? a.Mp : a.Np;   // owned dimension
    const int nsplit = a.nsplit;
    const int split = blockIdx.x % nsplit;
    const int qblk = blockIdx.x / nsplit;
    int q0 = (qblk * 4 + wave) * 32;     // wave-uniform (SGPR)
    const bool active = q0 < Q;
    if (!active) q0 = Q - 32;
    const size_t pair = CHECK ? (size_t)blockIdx.y : 0;   // the check of a batched solver: one launch over all its pairs (FusedArgs::strideW)
    const float *__restrict__ V = (WSTEP ? a.H : a.W) + pair * (WSTEP ? a.strideH : a.strideW);
    const float *__restrict__ U = (WSTEP ? a.W : a.H) + pair * (WSTEP ? a.strideW : a.strideH);
    const long ldv = WSTEP ? a.Kp : a.Mp, ldu = WSTEP ? a.Mp : a.Kp, ldx = a.Mp;
    const int nchunks = P / 32;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * cps;
    const int c_end = (c_begin + cps < nchunks) ? (c_begin + cps) : nchunks;

    float ub[KT * 16];
    load_u_split<KT, WSTEP>(ub, U, ldu, q0, c, h);

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    double kl = 0.0, dabs = 0.0, xabs = 0.0;
    if (c_begin < c_end) {
        // ---- per-thread constants: 32-bit lane offsets (floats) from wave-uniform chunk bases
        // V staging: H-step rows k = (tid>>3) + 32q, 16-B piece i4 = tid&7; W-step column i = (tid>>3)&31, piece k4 = 8q + (tid&7)
        // (byte offsets: "uniform pointer + zext(32-bit VGPR)" is the form hipcc turns into global_load ... saddr,
        //  i.e. no per-load 64-bit VALU address arithmetic)
        const unsigned voff0 = 4u * (WSTEP ? (unsigned)(4 * (tid & 7)) + (unsigned)((tid >> 3) & 31) * (unsigned)ldv
                                           : (unsigned)(4 * (tid & 7)) + (unsigned)(tid >> 3) * (unsigned)ldv);
        const unsigned vstep = 4u * (WSTEP ? 32u : 32u * (unsigned)ldv);              // bytes per q (32-bit on purpose)
        const size_t vchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldv : (size_t)32);   // bytes per chunk
        // X tile: 32 rows of 128 B; row = (lane>>3) + 8i at stride ldx, 16-B piece lane&7
        const unsigned xoff0 = 4u * ((unsigned)(4 * (lane & 7)) + (unsigned)(lane >> 3) * (unsigned)ldx);
        const unsigned xstep = 4u * 8u * (unsigned)ldx;                              // bytes per i (32-bit on purpose)
        const char *__restrict__ xbase = reinterpret_cast<const char *>(WSTEP ? a.X + (size_t)q0 : a.X + (size_t)q0 * (size_t)ldx);
        const size_t xchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldx : (size_t)32);
        // (no __restrict__ here: the patch is written and read back through these pointers within one wave)
        float *xt = smem + 2 * VBUF + wave * kXtFloats;                       // this wave's X patch
        float *xt_w = xt + (lane >> 3) * kXtLd + 4 * (lane & 7);              // write position (+ 8i rows)
        const float *xt_r = WSTEP ? xt + 4 * h * kXtLd + c                    // + rho(r) rows
                                  : xt + c * kXtLd + 4 * h;                   // + 8g floats
        // LDS operand bases (floats) inside a V buffer
        const int p1_off = N1 * h * kLdv + c;     // product 1, step ss: k = ss + N1*h  ->  + ss*kLdv
        const int p2_off = c * kLdv + 4 * h;      // product 2: + 32*t*kLdv + rho(r)

        f32x4 st[KT];
        f32x4 xg[4];
        float xr[16];
        // one global load per call, so that the loop can space them out between MFMAs: a burst of 1-KiB loads from
        // all four waves saturates the CU's ~70 B/clk vector-memory path and stalls every wave in issue (~730 cycles
        // per chunk measured); one load per several MFMAs costs ~22 cycles each.
        // The lane offsets are made opaque per chunk: otherwise their zero-extension is hoisted out of the loop and
        // each load needs a v_lshl_add_u64 instead of the global_load ... v_off32, s[base] form.
        unsigned vo = voff0, xo = xoff0;
        const char *__restrict__ vcur = reinterpret_cast<const char *>(V);
        const char *__restrict__ xcur = xbase;
        auto set_chunk = [&](int ch) {
            vo = voff0; xo = xoff0;
            asm volatile("" : "+v"(vo), "+v"(xo));
            vcur = reinterpret_cast<const char *>(V) + (size_t)ch * vchunk;
            xcur = xbase + (size_t)ch * xchunk;
        };
        auto stage_load_one = [&](int q) { st[q] = *reinterpret_cast<const f32x4 *>((vcur + (size_t)q * (size_t)vstep) + vo); };
        auto x_load_one = [&](int i) { xg[i] = *reinterpret_cast<const f32x4 *>((xcur + (size_t)i * (size_t)xstep) + xo); };
        // one 4-byte LDS write of the staged chunk (piece w of 4*KT per thread)
        auto stage_store_one = [&](float *__restrict__ vl, int w) {
            const int q = w / 4, cc = w % 4;
            if (!WSTEP) { const int k = (tid >> 3) + 32 * q, i4 = tid & 7; vl[k * kLdv + 4 * i4 + cc] = st[q][cc]; }
            else        { const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31; vl[(4 * k4 + cc) * kLdv + i] = st[q][cc]; }
        };
        auto x_relayout = [&]() {   // xg (coalesced layout) -> LDS patch -> xr (accumulator layout); same wave, DS ops are in order
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4 *>(xt_w + 8 * i * kXtLd) = xg[i];
            if (!WSTEP) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xt_r + 8 * g);
                    xr[4 * g] = v[0]; xr[4 * g + 1] = v[1]; xr[4 * g + 2] = v[2]; xr[4 * g + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) xr[r] = xt_r[rho(r) * kXtLd];
            }
        };

        unsigned long long tk0 = 0, tk1 = 0, tk2 = 0, tk3 = 0, tk4 = 0, tk5 = 0, seg[7] = {0, 0, 0, 0, 0, 0, 0};
        set_chunk(c_begin);
#pragma unroll
        for (int q = 0; q < KT; ++q) stage_load_one(q);
#pragma unroll
        for (int i = 0; i < 4; ++i) x_load_one(i);
        stage_store<KT, WSTEP>(st, smem, tid);
        x_relayout();
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            NMF_STAMP(tk0);
            const int par = (ch - c_begin) & 1;
            const float *__restrict__ vb = smem + par * VBUF;
            float *__restrict__ vn = smem + (par ^ 1) * VBUF;
            // Branch-free body: past the last chunk the "next chunk" is the current one again (its image lands in the
            // other LDS buffer and in registers nobody reads).
            const int chn = (ch + 1 < c_end) ? ch + 1 : ch;
            set_chunk(chn);
            // ---- product 1: one dependent chain, operands through a ring of single ds_read_b32
            const lds_float *b1 = (const lds_float *)vb + p1_off;
            float ar[D];
#pragma unroll
            for (int i = 0; i < D; ++i) ar[i] = lds_ld(b1 + i * kLdv);
            NMF_STAMP(tk1);
            // S accumulates in VGPRs (inline asm, "v" constraint): the divide reads it without 16
            // v_accvgpr_read, and hipcc stops parking an accumulator tile elsewhere to reuse its AGPRs.
            // hipcc pads nothing around asm: the s_nop run below covers MFMA-result -> VALU-read.
            f32x16 s;
            constexpr int NLOAD = KT + 4;   // 4 X pieces first, then KT pieces of V, spread evenly over the chain
#pragma unroll
            for (int ss = 0; ss < N1; ++ss) {
                if (ss == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=v"(s) : "v"(ar[0]), "v"(ub[0]));
                else         asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(s) : "v"(ar[ss % D]), "v"(ub[ss]));
                if (ss + D < N1) ar[ss % D] = lds_ld(b1 + (ss + D) * kLdv);
                constexpr int G = N1 / (NLOAD + 1);            // one load every G MFMAs
                if (ss >= G && ss % G == 0 && ss / G - 1 < NLOAD) {
                    const int j = ss / G - 1;
                    if (j < 4) x_load_one(j); else stage_load_one(j - 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s));
            NMF_STAMP(tk2);
            if (CHECK) {
                float fkl = 0.f, fd = 0.f, fx = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float x = xr[r], y = clamp_eps(s[r]);
                    // sum x log y and sum |x - y|; the X-only terms of cuda/matrix.cu:592,517-518 are summed once at upload and
                    // sum y comes from the normalisers (launch_check_compose).  Padding: x = 0, y = EPS, both terms vanish to 1e-16
                    fkl = __builtin_fmaf(x, log2_hw(y), fkl);
                    fd += fabsf(x - y);
                }
                kl += (double)fkl; dabs += (double)fd; xabs += (double)fx;
                x_relayout();
#pragma unroll
                for (int w = 0; w < 4 * KT; ++w) stage_store_one(vn, w);
                __syncthreads();
                continue;
            }
            // ---- first operands of product 2 (LDS, hidden) before the VALU block
            const lds_float *b2 = (const lds_float *)vb + p2_off;
            float a2[D];
#pragma unroll
            for (int e = 0; e < D; ++e) a2[e] = lds_ld(b2 + 32 * (e % KT) * kLdv + rho(e / KT));
            // ---- quotient, all 16 rows in ONE block of VALU work (set_epsilon + vec_div, cuda/nmf.cu:128-131)
            // (sched_barrier: instruction selection otherwise sinks each divide next to the MFMA row that uses
            //  it, and every VALU<->MFMA switch costs ~13 cycles on top of the VALU issue time)
            float z[16];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 16; ++r) z[r] = quotient<DIV>(xr[r], clamp_eps(s[r]));
            __builtin_amdgcn_sched_barrier(0);
            // ---- next chunk's X tile into the accumulator layout (LDS only)
            x_relayout();
            NMF_STAMP(tk3);
            // ---- product 2: KT independent accumulators, no VALU inside
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int t = 0; t < KT; ++t) {
                    const int e = r * KT + t;
                    acc[t] = NMF_MFMA(a2[e % D], z[r], acc[t]);
                    if (e + D < 16 * KT) a2[e % D] = lds_ld(b2 + 32 * ((e + D) % KT) * kLdv + rho((e + D) / KT));
                    // the next chunk's LDS image: 4*KT single writes, one every 2nd step from step E/8 (a burst of them
                    // saturates the ~75 B/clk LDS store path and delays the operand reads queued behind it)
                    constexpr int E0 = (16 * KT) / 8;
                    if (e >= E0 && (e - E0) % 2 == 0 && (e - E0) / 2 < 4 * KT) {
                        stage_store_one(vn, (e - E0) / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            NMF_STAMP(tk4);
            __syncthreads();
            NMF_STAMP(tk5);
            if (STAMP) { seg[0] += tk1 - tk0; seg[1] += tk2 - tk1; seg[2] += tk3 - tk2; seg[3] += tk4 - tk3; seg[4] += tk5 - tk4; }
        }
        if (STAMP && lane == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(a.partials) + ((size_t)blockIdx.x * 4 + wave) * 7;
#pragma unroll
            for (int i = 0; i < 7; ++i) dbg[i] = seg[i];
        }
    }
    if (CHECK) {
        if (!active) { kl = 0.0; dabs = 0.0; xabs = 0.0; }
        block_reduce3(kl, dabs, xabs, chk_part + 3 * ((size_t)blockIdx.x + pair * gridDim.x), tid);
        return;
    }
    if (!active) return;
    fused_epilogue<KT, WSTEP, PARTIAL>(a, acc, split, q0, c, h, ldu);
}

template <int KT>
static hipError_t launch_fused_kt(const FusedArgs &a, bool wstep, hipStream_t stream) {
    const int Q = wstep ? a.Mp : a.Np;
    const int nqblk = (Q + 127) / 128;
    const dim3 grid((unsigned)(nqblk * a.nsplit)), block(256);
    const bool partial = a.partial != 0;
    int variant = fused_variant();
    if (variant == 0) variant = 3;
    // v3 addresses the streamed factor and the X tile with 32-bit lane offsets
    if (variant == 3 && ((size_t)a.Kp * (size_t)a.Mp >= ((size_t)1 << 31) || (size_t)40 * (size_t)a.Mp >= ((size_t)1 << 31))) variant = 1;
    const size_t vbuf = (size_t)KT * 32 * kLdv * sizeof(float);
    const size_t lds = variant == 1 ? 2 * vbuf : 2 * vbuf + 4 * kXtFloats * sizeof(float);
#define NMF_LAUNCH_FUSED(...)                                                                             \
    do {                                                                                                  \
        {                                                                                                 \
            hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                            \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        note_kernel((const void *)__VA_ARGS__, stream); \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a);                                   \
    } while (0)
#define NMF_LAUNCH_FUSED3(...)                                                                            \
    do {                                                                                                  \
        {                                                                                                 \
            hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                            \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        note_kernel((const void *)__VA_ARGS__, stream); \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a, (double *)nullptr);                \
    } while (0)
    if (variant == 1) {
        if (!wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, false, false>);
        else if (!wstep && partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, false, true>);
        else if (wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, true, false>);
        else NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, true, true>);
    } else {
        if (!wstep && !partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, false, false, 0>);
        else if (!wstep && partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, false, true, 0>);
        else if (wstep && !partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, true, false, 0>);
        else NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, true, true, 0>);
    }
#undef NMF_LAUNCH_FUSED
#undef NMF_LAUNCH_FUSED3
    return hipGetLastError();
}

#ifdef NMF_DIAGNOSTICS   // make DIAG=1: stamp and ablation instantiations, reached through nmf_solver_time_piece(which >= 100)
// diagnostic: v3 H-step (KT = 8, in place, IEEE divide) with in-kernel stamps; a.partials receives 5 x uint64 per wave
hipError_t launch_fused_stamp(const FusedArgs &a, hipStream_t stream) {
    if (a.Kp != 256 || !a.partials) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.Np + 127) / 128)), block(256);
    const size_t lds = (size_t)2 * 8 * 32 * kLdv * sizeof(float) + 4 * kXtFloats * sizeof(float);
    (void)hipFuncSetAttribute((const void *)fused_step_kernel_v3<8, false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    note_kernel((const void *)fused_step_kernel_v3<8, false, false, 0, true>, stream);

    hipLaunchKernelGGL((fused_step_kernel_v3<8, false, false, 0, true>), grid, block, lds, stream, a, (double *)nullptr);
    return hipGetLastError();
}

// timing probe: v1 H-step kernel (KT = 8, in place) with an ablation mask
hipError_t launch_fused_probe(const FusedArgs &a, int abl, hipStream_t stream) {
    if (a.Kp != 256) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.Np + 127) / 128)), block(256);
    const size_t lds = (size_t)2 * 8 * 32 * kLdv * sizeof(float);
#define NMF_PROBE(A_)                                                                                                   \
    case A_:                                                                                                            \
        (void)hipFuncSetAttribute((const void *)fused_step_kernel_v1<8, false, false, A_>,                              \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                \
        hipLaunchKernelGGL((fused_step_kernel_v1<8, false, false, A_>), grid, block, lds, stream, a);                   \
        break;
    switch (abl) {
        NMF_PROBE(0) NMF_PROBE(1) NMF_PROBE(2) NMF_PROBE(3) NMF_PROBE(4) NMF_PROBE(5) NMF_PROBE(6) NMF_PROBE(7)
        NMF_PROBE(8) NMF_PROBE(16) NMF_PROBE(32) NMF_PROBE(24) NMF_PROBE(40) NMF_PROBE(48) NMF_PROBE(11) NMF_PROBE(19) NMF_PROBE(35) NMF_PROBE(71)
        default: return hipErrorInvalidValue;
    }
#undef NMF_PROBE
    return hipGetLastError();
}

#endif   // NMF_DIAGNOSTICS

hipError_t launch_fused32(const FusedArgs &a, bool wstep, hipStream_t stream) {
    switch (a.Kp / 32) {
        case 1: return launch_fused_kt<1>(a, wstep, stream);
        case 2: return launch_fused_kt<2>(a, wstep, stream);
        case 4: return launch_fused_kt<4>(a, wstep, stream);
        case 8: return launch_fused_kt<8>(a, wstep, stream);
        default: return hipErrorInvalidValue;
    }
}

// =====================================================================================
// Convergence check: KL(X || WH), sum|X - WH|, sum|X|  (reduce1d_div / reduce1d_diff,
// cuda/matrix.cu:505-640) fused behind product 1 so W*H is never materialised.
// =====================================================================================
template <int KT>
__global__ __launch_bounds__(256, 1) void check_kernel(const float *__restrict__ W, const float *__restrict__ H, const float *__restrict__ X,
                                                       int Mp, int Np, int Kp, double *__restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VBUF = KT * 32 * kLdv;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, c = lane & 31, h = lane >> 5;
    int q0 = (blockIdx.x * 4 + wave) * 32;
    const bool active = q0 < Np;
    if (!active) q0 = Np - 32;
    float ub[KT * 16];
    load_u<KT, false>(ub, H, Kp, q0, c, h);
    double kl = 0.0, dabs = 0.0, xabs = 0.0;
    const int nchunks = Mp / 32;
    f32x4 st[KT];
    float xr[16];
    stage_load<KT, false>(st, W, Mp, 0, tid);
    load_x<false>(xr, X, Mp, 0, q0, c, h);
    stage_store<KT, false>(st, smem, tid);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ++ch) {
        const float *__restrict__ vb = smem + (ch & 1) * VBUF;
        float *__restrict__ vn = smem + ((ch & 1) ^ 1) * VBUF;
        const bool more = ch + 1 < nchunks;
        if (more) stage_load<KT, false>(st, W, Mp, (ch + 1) * 32, tid);
        const f32x16 s = product1<KT>(ub, vb, c, h);
        float fkl = 0.f, fd = 0.f, fx = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float x = xr[r], y = clamp_eps(s[r]);
            fkl = __builtin_fmaf(x, log2_hw(y), fkl);   // see fused_step_kernel_v3<CHECK>
            fd += fabsf(x - y);
        }
        kl += (double)fkl; dabs += (double)fd; xabs += (double)fx;
        if (more) load_x<false>(xr, X, Mp, (ch + 1) * 32, q0, c, h);
        if (more) stage_store<KT, false>(st, vn, tid);
        __syncthreads();
    }
    if (!active) { kl = 0.0; dabs = 0.0; xabs = 0.0; }
    block_reduce3(kl, dabs, xabs, part + 3 * (size_t)blockIdx.x, tid);
}

template <int KT>
static hipError_t launch_check_kt(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream,
                                  int batch, size_t strideW, size_t strideH) {
    // the production half-step kernel in CHECK mode (product 1 + KL terms); NMF_FUSED_VARIANT=1 keeps the first-generation check_kernel
    if (fused_variant() != 1 && (size_t)Kp * (size_t)Mp < ((size_t)1 << 31)) {
        FusedArgs a;
        a.W = W; a.H = H; a.X = X; a.U_out = nullptr; a.partials = nullptr; a.norm = nullptr;
        a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.nsplit = 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
        a.strideW = strideW; a.strideH = strideH;
        const size_t lds3 = (size_t)2 * KT * 32 * kLdv * sizeof(float) + 4 * kXtFloats * sizeof(float);
        {
            hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_v3<KT, false, false, 0, false, true>, lds3);
            if (e != hipSuccess) return e;
        }
        note_kernel((const void *)fused_step_kernel_v3<KT, false, false, 0, false, true>, stream);

        hipLaunchKernelGGL((fused_step_kernel_v3<KT, false, false, 0, false, true>), dim3(check_num_groups(Np, Kp), (unsigned)batch), dim3(256), lds3, stream, a, part);
        return hipGetLastError();
    }
    const size_t lds = (size_t)2 * KT * 32 * kLdv * sizeof(float);
    {
        hipError_t e = ensure_dynamic_lds((const void *)check_kernel<KT>, lds);
        if (e != hipSuccess) return e;
    }
    note_kernel((const void *)check_kernel<KT>, stream);
    for (int b = 0; b < batch; ++b) {   // the first-generation check kernel takes one pair at a time
        hipLaunchKernelGGL((check_kernel<KT>), dim3(check_num_groups(Np, Kp)), dim3(256), lds, stream, W + (size_t)b * strideW, H + (size_t)b * strideH, X, Mp, Np, Kp,
                           part + 3 * (size_t)check_num_groups(Np, Kp) * b);
    }
    return hipGetLastError();
}

hipError_t launch_check32(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream,
                          int batch, size_t strideW, size_t strideH) {
    switch (Kp / 32) {
        case 1: return launch_check_kt<1>(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH);
        case 2: return launch_check_kt<2>(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH);
        case 4: return launch_check_kt<4>(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH);
        case 8: return launch_check_kt<8>(W, H, X, Mp, Np, Kp, part, stream, batch, strideW, strideH);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace nmf
