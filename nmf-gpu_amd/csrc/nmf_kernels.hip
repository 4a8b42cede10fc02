// nmf_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the update_div hot path.
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
#include "nmf_kernels.h"

#include <cstdlib>
#include <mutex>
#include <set>
#include <utility>

namespace nmf {

// Kernels that need more than 64 KiB of dynamic LDS must opt in once per (kernel, device).
static hipError_t ensure_dynamic_lds(const void *fn, size_t bytes) {
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

// =====================================================================================
// Fused half-step
// =====================================================================================
// LDS image of one streamed chunk: Vl[k][p], p = 0..31 within the chunk, row stride 33 floats.
//   product-1 A operand  Vl[(2s+h)*33 + c]          : 32 consecutive banks            -> conflict-free
//   product-2 A operand  Vl[(32t+c)*33 + rho(r)+4h] : stride 33 (odd) across 32 lanes -> conflict-free
constexpr int kLdv = 33;

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

__device__ __forceinline__ void block_reduce3(double v0, double v1, double v2, double *out3, int tid);   // defined with the check kernels below

// LLVM SchedGroupMask bits for __builtin_amdgcn_sched_group_barrier
#define NMF_SG_VALU 0x002
#define NMF_SG_MFMA 0x008
#define NMF_SG_VMEM_READ 0x020
#define NMF_SG_DS_READ 0x100
#define NMF_SG_DS_WRITE 0x200

// Product 1: S(32 p x 32 q) = V_chunk * U_slice, one dependent chain of KT*16 MFMAs (the 32x32x2 f32
// MFMA has issue interval = dependent latency = 64 cycles, so a single chain runs at full rate as
// long as its A operand is already in a register).  The A operands come from LDS through a ring of
// kRing registers loaded kRing MFMAs (>= 512 cycles) ahead of their use; hipcc otherwise emits
// ds_read -> s_waitcnt lgkmcnt(0) -> MFMA and exposes the LDS latency on every pair.
constexpr int kRing = 8;
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
// Production fused half-step: software-pipelined across chunks.
// In the steady state one loop iteration issues, interleaved in this order,
//   * product 2 of chunk j   (16 rows x KT independent MFMAs, operands from LDS buffer j),
//   * product 1 of chunk j+1 (one dependent chain of 16*KT MFMAs, LDS buffer j+1), finished two
//     rows early so its result is back before the first divide that needs it,
//   * the IEEE divide of the next row of Z, cut into four stages placed between MFMAs,
//   * the LDS writes of chunk j+2 (global loads issued at the top of the iteration),
// so the matrix pipe never waits for an LDS read, a divide or a chunk boundary.  Three LDS buffers,
// one barrier per chunk.  The translation unit is compiled with the machine scheduler off: the
// statement order below IS the issue order.
// =====================================================================================
struct DivPipe {   // state of one correctly-rounded fp32 division x / y, identical to hipcc's expansion of `/`
    float x, y, ds, rc, ns, q;
    bool fl;
};
__device__ __forceinline__ void div_stage(int stage, DivPipe &d) {
    if (stage == 0) {
        bool unused;
        d.ds = __builtin_amdgcn_div_scalef(d.x, d.y, false, &unused);
        d.rc = __builtin_amdgcn_rcpf(d.ds);
    } else if (stage == 1) {
        const float e0 = __builtin_fmaf(-d.ds, d.rc, 1.0f);
        d.rc = __builtin_fmaf(e0, d.rc, d.rc);
        d.ns = __builtin_amdgcn_div_scalef(d.x, d.y, true, &d.fl);
    } else if (stage == 2) {
        d.q = d.ns * d.rc;
        const float e1 = __builtin_fmaf(-d.ds, d.q, d.ns);
        d.q = __builtin_fmaf(e1, d.rc, d.q);
    } else {
        const float e2 = __builtin_fmaf(-d.ds, d.q, d.ns);
        const float r = __builtin_amdgcn_div_fmasf(e2, d.rc, d.q, d.fl);
        d.q = __builtin_amdgcn_div_fixupf(r, d.y, d.x);
    }
}

// number of product-1 steps issued once product-2 steps 0..e have been issued
template <int KT>
__device__ __forceinline__ constexpr int p1_cum(int e) {
    constexpr int N = 16 * KT, END = N - 2 * KT;
    if (e < 0) return 0;
    const int v = ((e + 1) * N + END - 1) / END;
    return v < N ? v : N;
}

template <int KT, bool WSTEP, bool NEXT, bool STORE>
__device__ __forceinline__ void pipelined_chunk(f32x16 (&acc)[KT], const float (&ub)[KT * 16], const float *__restrict__ vb_cur,
                                                const float *__restrict__ vb_nxt, float *__restrict__ vb_st, const f32x16 &s_cur,
                                                f32x16 &s_nxt, const float (&x_cur)[16], const float (&x_nxt)[16], float (&a2)[kRing],
                                                float &zc, const f32x4 (&st)[KT], int tid, int c, int h) {
    constexpr int N = 16 * KT;
    static_assert(N % kRing == 0, "ring must divide the step count");
    const float *__restrict__ base1 = vb_nxt + h * kLdv + c;        // product-1 operand of step i : base1[2*i*kLdv]
    const float *__restrict__ base2 = vb_cur + c * kLdv + 4 * h;    // product-2 operand (r,t)     : base2[32*t*kLdv + rho(r)]
    const float *__restrict__ base2n = vb_nxt + c * kLdv + 4 * h;
    float a1[kRing];
    if (NEXT) {
#pragma unroll
        for (int i = 0; i < kRing; ++i) a1[i] = base1[2 * i * kLdv];
#pragma unroll
        for (int r = 0; r < 16; ++r) s_nxt[r] = 0.f;
    }
    DivPipe dv;
    dv.x = dv.y = dv.ds = dv.rc = dv.ns = dv.q = 0.f;
    dv.fl = false;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const bool have_div = (r + 1 < 16) || NEXT;   // row r+1 of this chunk, or row 0 of the next one
#pragma unroll
        for (int t = 0; t < KT; ++t) {
            const int e = r * KT + t;
            if (have_div) {
#pragma unroll
                for (int sg = 0; sg < 4; ++sg) {
                    if ((sg * KT) / 4 == t) {
                        if (sg == 0) {
                            if (r + 1 < 16) { dv.x = x_cur[r + 1]; dv.y = clamp_eps(s_cur[r + 1]); }
                            else            { dv.x = x_nxt[0];     dv.y = clamp_eps(s_nxt[0]); }
                        }
                        div_stage(sg, dv);
                    }
                }
            }
            // product 2
            acc[t] = NMF_MFMA(a2[e % kRing], zc, acc[t]);
            if (e + kRing < N) {
                a2[e % kRing] = base2[32 * ((e + kRing) % KT) * kLdv + rho((e + kRing) / KT)];
            } else if (NEXT) {   // refill the ring with the first operands of the next chunk's product 2
                const int en = e + kRing - N;
                a2[e % kRing] = base2n[32 * (en % KT) * kLdv + rho(en / KT)];
            }
            // product 1 of the next chunk
            if (NEXT) {
#pragma unroll
                for (int i = p1_cum<KT>(e - 1); i < p1_cum<KT>(e); ++i) {
                    s_nxt = NMF_MFMA(a1[i % kRing], ub[i], s_nxt);
                    if (i + kRing < N) a1[i % kRing] = base1[2 * (i + kRing) * kLdv];
                }
            }
            // LDS writes of chunk j+2 (4*KT per thread), one every 2nd slot of the second half of the
            // chunk: its global loads were issued at the top of the iteration and need ~2 us to land
            if (STORE && e >= N / 2 && ((e - N / 2) % 2) == 1) {
                const int w = (e - N / 2) / 2, q = w / 4, cc = w % 4;
                const int f = tid + q * 256;
                if (!WSTEP) {
                    const int k = f >> 3, i4 = f & 7;
                    vb_st[k * kLdv + 4 * i4 + cc] = st[q][cc];
                } else {
                    const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31;
                    vb_st[(4 * k4 + cc) * kLdv + i] = st[q][cc];
                }
            }
            // pin the issue order: without this fence hipcc's instruction selection clusters the
            // product-1 MFMAs and sinks the LDS reads next to their uses
            __builtin_amdgcn_sched_barrier(0);
        }
        if (have_div) zc = dv.q;
    }
}

template <int KT, bool WSTEP, bool PARTIAL>
__global__ __launch_bounds__(256, 1) void fused_step_kernel(FusedArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int VBUF = KT * 32 * kLdv;
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
    const int nch = c_end - c_begin;

    float ub[KT * 16];
    load_u<KT, WSTEP>(ub, U, ldu, q0, c, h);

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    if (nch > 0) {
        f32x4 st[KT];
        float x_cur[16], x_nxt[16];
        float *b0 = smem, *b1 = smem + VBUF, *b2 = smem + 2 * VBUF;
        // prologue: chunks 0 and 1 into LDS, product 1 of chunk 0 on its own
        stage_load<KT, WSTEP>(st, V, ldv, c_begin * 32, tid);
        load_x<WSTEP>(x_cur, a.X, ldx, c_begin * 32, q0, c, h);
        stage_store<KT, WSTEP>(st, b0, tid);
        if (nch > 1) {
            stage_load<KT, WSTEP>(st, V, ldv, (c_begin + 1) * 32, tid);
            load_x<WSTEP>(x_nxt, a.X, ldx, (c_begin + 1) * 32, q0, c, h);
            stage_store<KT, WSTEP>(st, b1, tid);
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) x_nxt[r] = 0.f;
        }
        __syncthreads();
        f32x16 s_cur = product1<KT>(ub, b0, c, h);
        f32x16 s_nxt;
        float a2[kRing];
        {
            const float *__restrict__ base2 = b0 + c * kLdv + 4 * h;
#pragma unroll
            for (int e = 0; e < kRing; ++e) a2[e] = base2[32 * (e % KT) * kLdv + rho(e / KT)];
        }
        float zc = x_cur[0] / clamp_eps(s_cur[0]);
        for (int j = 0; j + 1 < nch; ++j) {
            // Branch-free body (a second instantiation inside the loop makes the register allocator copy
            // every accumulator at the join).  Past the end the chunk index is clamped: the surplus
            // chunk lands in an LDS buffer / registers nobody reads again.
            const int cn = (j + 2 < nch) ? (c_begin + j + 2) : (c_end - 1);
            stage_load<KT, WSTEP>(st, V, ldv, cn * 32, tid);
            pipelined_chunk<KT, WSTEP, true, true>(acc, ub, b0, b1, b2, s_cur, s_nxt, x_cur, x_nxt, a2, zc, st, tid, c, h);
            s_cur = s_nxt;
#pragma unroll
            for (int r = 0; r < 16; ++r) x_cur[r] = x_nxt[r];
            __syncthreads();
            // after the barrier, so that its vmcnt(0) finds no load of ours in flight
            load_x<WSTEP>(x_nxt, a.X, ldx, cn * 32, q0, c, h);
            float *tmp = b0; b0 = b1; b1 = b2; b2 = tmp;
        }
        // last chunk: product 2 only
        pipelined_chunk<KT, WSTEP, false, false>(acc, ub, b0, b1, b2, s_cur, s_nxt, x_cur, x_nxt, a2, zc, st, tid, c, h);
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
constexpr int kXtLd = 36;                       // X patch row stride in floats (16-B aligned, b128 conflict-free)
constexpr int kXtFloats = 32 * kXtLd;           // per wave

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

// LDS pointer with its address space spelled out: a volatile load through a generic pointer is not
// rewritten by address-space inference and would become flat_load + 64-bit address arithmetic.
typedef __attribute__((address_space(3))) float lds_float;
__device__ __forceinline__ float lds_ld(const lds_float *p) { return *reinterpret_cast<const volatile lds_float *>(p); }

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
    const float *__restrict__ V = WSTEP ? a.H : a.W;
    const float *__restrict__ U = WSTEP ? a.W : a.H;
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
                    if (x > 0.f) {   // padding is exactly 0; real inputs are >= EPS (cuda/nmf.cu:211)
                        fkl += x * (logf(x) - logf(y)) - x + y;   // cuda/matrix.cu:592
                        fd += fabsf(x - y);                       // cuda/matrix.cu:517
                        fx += fabsf(x);                           // cuda/matrix.cu:518
                    }
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
        block_reduce3(kl, dabs, xabs, chk_part + 3 * (size_t)blockIdx.x, tid);
        return;
    }
    if (!active) return;
    fused_epilogue<KT, WSTEP, PARTIAL>(a, acc, split, q0, c, h, ldu);
}

// Variants (NMF_FUSED_VARIANT), for K <= 256: unset = production choice (16-column kernel at two workgroups per CU for
// K = 64/128/256, v3 for K = 32); 3 = v3 (32-column kernel) everywhere; 1 = first chunk-serial kernel; 2 = software-
// pipelined experiment.  NMF_FAST_DIVIDE=1 selects the refined-reciprocal quotient (<= 1 ulp) in variant 3.
static int fused_variant() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NMF_FUSED_VARIANT"); v = (e && e[0] >= '1' && e[0] <= '3') ? (e[0] - '0') : 0; }   // 0 = automatic choice
    return v;
}
static int fused_fast_divide() {
    static int v = -1;
    if (v < 0) { const char *e = getenv("NMF_FAST_DIVIDE"); v = (e && e[0] == '1') ? 1 : 0; }
    return v;
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
    const size_t lds = variant == 1 ? 2 * vbuf : variant == 2 ? 3 * vbuf : 2 * vbuf + 4 * kXtFloats * sizeof(float);
#define NMF_LAUNCH_FUSED(...)                                                                             \
    do {                                                                                                  \
        {                                                                                                 \
            hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                            \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a);                                   \
    } while (0)
#define NMF_LAUNCH_FUSED3(...)                                                                            \
    do {                                                                                                  \
        {                                                                                                 \
            hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                            \
            if (e != hipSuccess) return e;                                                                \
        }                                                                                                 \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a, (double *)nullptr);                \
    } while (0)
    if (variant == 1) {
        if (!wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, false, false>);
        else if (!wstep && partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, false, true>);
        else if (wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, true, false>);
        else NMF_LAUNCH_FUSED(fused_step_kernel_v1<KT, true, true>);
    } else if (variant == 2) {
        if (!wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel<KT, false, false>);
        else if (!wstep && partial) NMF_LAUNCH_FUSED(fused_step_kernel<KT, false, true>);
        else if (wstep && !partial) NMF_LAUNCH_FUSED(fused_step_kernel<KT, true, false>);
        else NMF_LAUNCH_FUSED(fused_step_kernel<KT, true, true>);
    } else if (fused_fast_divide() || a.fast_divide) {
        if (!wstep && !partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, false, false, 1>);
        else if (!wstep && partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, false, true, 1>);
        else if (wstep && !partial) NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, true, false, 1>);
        else NMF_LAUNCH_FUSED3(fused_step_kernel_v3<KT, true, true, 1>);
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

// diagnostic: v3 H-step (KT = 8, in place, IEEE divide) with in-kernel stamps; a.partials receives 5 x uint64 per wave
hipError_t launch_fused_stamp(const FusedArgs &a, hipStream_t stream) {
    if (a.Kp != 256 || !a.partials) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((a.Np + 127) / 128)), block(256);
    const size_t lds = (size_t)2 * 8 * 32 * kLdv * sizeof(float) + 4 * kXtFloats * sizeof(float);
    (void)hipFuncSetAttribute((const void *)fused_step_kernel_v3<8, false, false, 0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
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

// =====================================================================================
// Fused half-step for 256 < K <= 512 (BASELINE config 5 has R = 512): same algorithm as v3 on
// v_mfma_f32_16x16x4_f32 with 16 owned columns per wave, so that the K x 16 accumulator (16*NB
// registers) and the K B-operands of product 1 (16*NB registers) fit the register file, NB = K/64.
// Lane maps of the 16x16x4 form (lane l: j = l & 15, kq = l >> 4):
//     A operand = A[row j][k kq],  B operand = B[k kq][col j],  result reg r = D[4 kq + r][j]
// so register r of a finished 16x16 tile is the B operand of a step whose four k indices are the tile
// rows 4 kq + r: the quotient again feeds product 2 straight from the accumulator registers.
// Product 1 runs two interleaved chains (the two 16-row halves of the 32-row chunk): the 16x16x4 MFMA
// issues every 32 cycles but needs 40 between dependent ones.
// k index of product-1 step s in lane group kq: 64 (s >> 4) + 16 kq + (s & 15): per-lane contiguous runs
// of 16 (16-B loads of the owned factor) and, with 33-float LDS rows, 32 distinct banks per half-wave.
// CHECK = true turns the kernel into the KL / rel-L1 check (product 1 only), see check_kernel.
// =====================================================================================
typedef const __attribute__((address_space(1))) char *global_bytes;
#define NMF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
constexpr int kXt16Floats = 32 * 20;   // per-wave X patch: H-step 16 x 36, W-step 32 x 20 floats

template <int NB, bool WSTEP, bool PARTIAL, int DIV, bool CHECK = false, int OCC = 1>
__global__ __launch_bounds__(256, OCC) void fused_step_kernel_k16(FusedArgs a, double *__restrict__ chk_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 64 * NB;
    constexpr int VBUF = K * kLdv;
    constexpr int N1 = 16 * NB;      // product-1 steps per 16-row tile
    constexpr int NT = 4 * NB;       // 16 x 16 accumulator tiles
    constexpr int NST = 2 * NB;      // staged 16-B pieces per thread per chunk
    constexpr int D = kRing;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int P = WSTEP ? a.Np : a.Mp;
    const int Q = WSTEP ? a.Mp : a.Np;
    const int nsplit = a.nsplit;
    const bool x_in_range = a.x_in_range != 0;
    const int split = blockIdx.x % nsplit;
    const int qblk = blockIdx.x / nsplit;
    int q0 = (qblk * 4 + wave) * 16;
    const bool active = q0 < Q;
    if (!active) q0 = Q - 16;
    const float *__restrict__ V = WSTEP ? a.H : a.W;
    const float *__restrict__ U = WSTEP ? a.W : a.H;
    const long ldv = WSTEP ? a.Kp : a.Mp, ldu = WSTEP ? a.Mp : a.Kp, ldx = a.Mp;
    const int nchunks = P / 32;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * cps;
    const int c_end = (c_begin + cps < nchunks) ? (c_begin + cps) : nchunks;

    // B operands of product 1: ub[s] = U(k(s, kq), q0 + j)
    float ub[N1];
    if (!WSTEP) {
        const float *__restrict__ col = U + (size_t)(16 * kq) + (size_t)(q0 + j) * ldu;
#pragma unroll
        for (int sb = 0; sb < NB; ++sb)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 64 * sb + 4 * e4);
                ub[16 * sb + 4 * e4] = v[0]; ub[16 * sb + 4 * e4 + 1] = v[1]; ub[16 * sb + 4 * e4 + 2] = v[2]; ub[16 * sb + 4 * e4 + 3] = v[3];
            }
    } else {
#pragma unroll
        for (int s = 0; s < N1; ++s) ub[s] = U[(size_t)(q0 + j) + (size_t)(64 * (s >> 4) + 16 * kq + (s & 15)) * ldu];
    }

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    double kl = 0.0, dabs = 0.0, xabs = 0.0;

    if (c_begin < c_end) {
        const unsigned voff0 = 4u * (WSTEP ? (unsigned)(4 * (tid & 7)) + (unsigned)((tid >> 3) & 31) * (unsigned)ldv
                                           : (unsigned)(4 * (tid & 7)) + (unsigned)(tid >> 3) * (unsigned)ldv);
        const unsigned vstep = 4u * (WSTEP ? 32u : 32u * (unsigned)ldv);
        const size_t vchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldv : (size_t)32);
        // X tile (32 p x 16 q): H-step 16 columns of 128 B (8 lanes per column), W-step 32 rows of 64 B (4 lanes per row)
        const unsigned xoff0 = WSTEP ? 4u * ((unsigned)(4 * (lane & 3)) + (unsigned)(lane >> 2) * (unsigned)ldx)
                                     : 4u * ((unsigned)(4 * (lane & 7)) + (unsigned)(lane >> 3) * (unsigned)ldx);
        const unsigned xstep = WSTEP ? 4u * 16u * (unsigned)ldx : 4u * 8u * (unsigned)ldx;
        const char *__restrict__ xbase = reinterpret_cast<const char *>(WSTEP ? a.X + (size_t)q0 : a.X + (size_t)q0 * (size_t)ldx);
        const size_t xchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldx : (size_t)32);
        float *xt = smem + 2 * VBUF + wave * kXt16Floats;   // no __restrict__: written and read back within the wave
        float *xt_w = WSTEP ? xt + (lane >> 2) * 20 + 4 * (lane & 3) : xt + (lane >> 3) * kXtLd + 4 * (lane & 7);
        const float *xt_r = WSTEP ? xt + 4 * kq * 20 + j : xt + j * kXtLd + 4 * kq;
        const int p1_off = 16 * kq * kLdv + j;     // + (64 (s>>4) + (s&15)) * kLdv + 16 T
        const int p2_off = j * kLdv + 4 * kq;      // + 16 t * kLdv + 16 T + r

        f32x4 st[NST];
        f32x4 xg[2];
        float xr[8];
        unsigned vo = voff0, xo = xoff0;
        const char *__restrict__ vcur = reinterpret_cast<const char *>(V);
        const char *__restrict__ xcur = xbase;
        auto set_chunk = [&](int ch) {
            vo = voff0; xo = xoff0;
            asm volatile("" : "+v"(vo), "+v"(xo));
            vcur = reinterpret_cast<const char *>(V) + (size_t)ch * vchunk;
            xcur = xbase + (size_t)ch * xchunk;
        };
        // the uniform part of every address is pinned in an SGPR pair (scalar adds are free next to the MFMAs;
        // left alone the compiler chains 64-bit VALU adds through the per-lane address instead)
        auto stage_load_one = [&](int q) {
            global_bytes base = (global_bytes)(vcur + (size_t)q * (size_t)vstep);
            asm volatile("" : "+s"(base));
            st[q] = *(const __attribute__((address_space(1))) f32x4 *)(base + vo);
        };
        auto x_load_one = [&](int i) {
            global_bytes base = (global_bytes)(xcur + (size_t)i * (size_t)xstep);
            asm volatile("" : "+s"(base));
            xg[i] = *(const __attribute__((address_space(1))) f32x4 *)(base + xo);
        };
        auto stage_store_one = [&](float *__restrict__ vl, int w) {
            const int q = w / 4, cc = w % 4;
            if (!WSTEP) { const int k = (tid >> 3) + 32 * q, i4 = tid & 7; vl[k * kLdv + 4 * i4 + cc] = st[q][cc]; }
            else        { const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31; vl[(4 * k4 + cc) * kLdv + i] = st[q][cc]; }
        };
        auto x_relayout = [&]() {
            if (!WSTEP) {
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(xt_w + 8 * i * kXtLd) = xg[i];
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(xt_r + 16 * T);
                    xr[4 * T] = v[0]; xr[4 * T + 1] = v[1]; xr[4 * T + 2] = v[2]; xr[4 * T + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(xt_w + 16 * i * 20) = xg[i];
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int r = 0; r < 4; ++r) xr[4 * T + r] = xt_r[(16 * T + r) * 20];
            }
        };

        set_chunk(c_begin);
#pragma unroll
        for (int q = 0; q < NST; ++q) stage_load_one(q);
#pragma unroll
        for (int i = 0; i < 2; ++i) x_load_one(i);
#pragma unroll
        for (int w = 0; w < 4 * NST; ++w) stage_store_one(smem, w);
        x_relayout();
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int par = (ch - c_begin) & 1;
            const float *__restrict__ vb = smem + par * VBUF;
            float *__restrict__ vn = smem + (par ^ 1) * VBUF;
            const int chn = (ch + 1 < c_end) ? ch + 1 : ch;
            set_chunk(chn);
            // ---- product 1: two interleaved chains, step index e = 2 s + T
            const lds_float *b1 = (const lds_float *)vb + p1_off;
            constexpr int E1 = 2 * N1;
            float ar[D];
#pragma unroll
            for (int e = 0; e < D; ++e) ar[e] = lds_ld(b1 + (64 * ((e >> 1) >> 4) + ((e >> 1) & 15)) * kLdv + 16 * (e & 1));
            if (OCC > 1) __builtin_amdgcn_s_setprio(1);
            f32x4 s0, s1;
            constexpr int NLOAD = NST + 2;
            constexpr int G = E1 / (NLOAD + 1);
#pragma unroll
            for (int e = 0; e < E1; ++e) {
                const int s = e >> 1;
                if (e == 0)      asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(s0) : "v"(ar[0]), "v"(ub[0]));
                else if (e == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(s1) : "v"(ar[1]), "v"(ub[0]));
                else if (e & 1)  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(ar[e % D]), "v"(ub[s]));
                else             asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(ar[e % D]), "v"(ub[s]));
                if (e + D < E1) {
                    const int en = e + D;
                    ar[e % D] = lds_ld(b1 + (64 * ((en >> 1) >> 4) + ((en >> 1) & 15)) * kLdv + 16 * (en & 1));
                }
                if (e >= G && e % G == 0 && e / G - 1 < NLOAD) {
                    const int l = e / G - 1;
                    if (l < 2) x_load_one(l); else stage_load_one(l - 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            asm volatile("s_nop 15\n\ts_nop 3" : "+v"(s0), "+v"(s1));
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            if (CHECK) {
                float fkl = 0.f, fd = 0.f, fx = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float x = xr[r], y = clamp_eps(r < 4 ? s0[r] : s1[r - 4]);
                    if (x > 0.f) { fkl += x * (logf(x) - logf(y)) - x + y; fd += fabsf(x - y); fx += fabsf(x); }
                }
                kl += (double)fkl; dabs += (double)fd; xabs += (double)fx;
                x_relayout();
#pragma unroll
                for (int w = 0; w < 4 * NST; ++w) stage_store_one(vn, w);
                __syncthreads();
                continue;
            }
            // ---- first operands of product 2, then the quotient in one VALU block
            const lds_float *b2 = (const lds_float *)vb + p2_off;
            constexpr int E2 = 8 * NT;   // order: (T, r) outer, tile t inner
            float a2[D];
#pragma unroll
            for (int e = 0; e < D; ++e) a2[e] = lds_ld(b2 + 16 * (e % NT) * kLdv + 16 * ((e / NT) >> 2) + ((e / NT) & 3));
            float z[8];
            __builtin_amdgcn_sched_barrier(0);
            quotient8<DIV>(xr, s0, s1, z, x_in_range);
            __builtin_amdgcn_sched_barrier(0);
            x_relayout();
            if (OCC > 1) __builtin_amdgcn_s_setprio(1);
            // ---- product 2: NT independent accumulators
            constexpr int E0 = E2 / 8;
#pragma unroll
            for (int g = 0; g < 8; ++g) {        // g = 4 T + r
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int e = g * NT + t;
                    acc[t] = NMF_MFMA16(a2[e % D], z[g], acc[t]);
                    if (e + D < E2) {
                        const int en = e + D, gn = en / NT, tn = en % NT;
                        a2[e % D] = lds_ld(b2 + 16 * tn * kLdv + 16 * (gn >> 2) + (gn & 3));
                    }
                    if (e >= E0 && (e - E0) % 2 == 0 && (e - E0) / 2 < 4 * NST) {
                        stage_store_one(vn, (e - E0) / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
    }
    if (CHECK) {
        if (!active) { kl = 0.0; dabs = 0.0; xabs = 0.0; }
        block_reduce3(kl, dabs, xabs, chk_part + 3 * (size_t)blockIdx.x, tid);
        return;
    }
    if (!active) return;
    // epilogue: lane holds Acc(k = 16 t + 4 kq + r, q0 + j)
    if (PARTIAL) {
        const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
        float *__restrict__ out = a.partials + (size_t)split * slab;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)(16 * t + 4 * kq) + (size_t)(q0 + j) * ldu) = acc[t];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(16 * t + 4 * kq + r) * ldu] = acc[t][r];
            }
        }
    } else {
        float *__restrict__ Uo = a.U_out;
        const float *__restrict__ nrm = a.norm;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = 16 * t + 4 * kq;
            if (!WSTEP) {
                float *p = Uo + (size_t)k + (size_t)(q0 + j) * ldu;
                f32x4 u = *reinterpret_cast<const f32x4 *>(p);
                const f32x4 n4 = *reinterpret_cast<const f32x4 *>(nrm + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = u[e] * (acc[t][e] / n4[e]);
                *reinterpret_cast<f32x4 *>(p) = u;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *p = Uo + (size_t)(q0 + j) + (size_t)(k + r) * ldu;
                    *p = *p * (acc[t][r] / nrm[k + r]);
                }
            }
        }
    }
}

template <int NB, int OCC>
static hipError_t launch_fused_k16(const FusedArgs &a, bool wstep, hipStream_t stream) {
    const int Q = wstep ? a.Mp : a.Np;
    const dim3 grid((unsigned)(((Q + 63) / 64) * a.nsplit)), block(256);
    const size_t lds = (size_t)2 * 64 * NB * kLdv * sizeof(float) + 4 * kXt16Floats * sizeof(float);
    const bool partial = a.partial != 0;
    const bool fast = fused_fast_divide() || a.fast_divide;
#define NMF_LAUNCH_K16(...)                                                                               \
    do {                                                                                                  \
        hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                                \
        if (e != hipSuccess) return e;                                                                    \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a, (double *)nullptr);                \
    } while (0)
    if (fast) {
        if (!wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, false, false, 1, false, OCC>);
        else if (!wstep && partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, false, true, 1, false, OCC>);
        else if (wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, true, false, 1, false, OCC>);
        else NMF_LAUNCH_K16(fused_step_kernel_k16<NB, true, true, 1, false, OCC>);
    } else {
        if (!wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, false, false, 0, false, OCC>);
        else if (!wstep && partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, false, true, 0, false, OCC>);
        else if (wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<NB, true, false, 0, false, OCC>);
        else NMF_LAUNCH_K16(fused_step_kernel_k16<NB, true, true, 0, false, OCC>);
    }
#undef NMF_LAUNCH_K16
    return hipGetLastError();
}

template <int NB, int OCC>
static hipError_t launch_check_k16(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream) {
    FusedArgs a;
    a.W = W; a.H = H; a.X = X; a.U_out = nullptr; a.partials = nullptr; a.norm = nullptr;
    a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.nsplit = 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
    const size_t lds = (size_t)2 * 64 * NB * kLdv * sizeof(float) + 4 * kXt16Floats * sizeof(float);
    hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_k16<NB, false, false, 0, true, OCC>, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((fused_step_kernel_k16<NB, false, false, 0, true, OCC>), dim3((Np + 63) / 64), dim3(256), lds, stream, a, part);
    return hipGetLastError();
}

// which kernel family serves a padded K: the 16-column kernel for K = 64/128/256 (two workgroups per CU) and for
// 256 < K <= 512 (one), v3 for K = 32 or when NMF_FUSED_VARIANT asks for it
static bool use_k16(int Kp) { return Kp > 256 || (Kp >= 64 && fused_variant() == 0); }

hipError_t launch_fused_step(const FusedArgs &a, bool wstep, hipStream_t stream) {
    if ((a.Mp | a.Np | a.Kp) & 31) return hipErrorInvalidValue;
    if (a.nsplit < 1 || (a.nsplit > 1 && !a.partial)) return hipErrorInvalidValue;
    if (use_k16(a.Kp)) {
        if (a.Kp % 64) return hipErrorInvalidValue;
        switch (a.Kp / 64) {
            case 1: return launch_fused_k16<1, 2>(a, wstep, stream);
            case 2: return launch_fused_k16<2, 2>(a, wstep, stream);
            case 4: return launch_fused_k16<4, 2>(a, wstep, stream);
            case 5: return launch_fused_k16<5, 1>(a, wstep, stream);
            case 6: return launch_fused_k16<6, 1>(a, wstep, stream);
            case 7: return launch_fused_k16<7, 1>(a, wstep, stream);
            case 8: return launch_fused_k16<8, 1>(a, wstep, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (a.Kp / 32) {
        case 1: return launch_fused_kt<1>(a, wstep, stream);
        case 2: return launch_fused_kt<2>(a, wstep, stream);
        case 4: return launch_fused_kt<4>(a, wstep, stream);
        case 8: return launch_fused_kt<8>(a, wstep, stream);
        default: return hipErrorInvalidValue;
    }
}

int fused_cols_per_group(int Kp) { return use_k16(Kp) ? 64 : 128; }
int fused_pad_k(int K) {   // padded K the fused kernels are instantiated for; 0 = not supported
    const int k32 = pad32(K);
    if (k32 <= 256) { int kt = k32 / 32, p = 1; while (p < kt) p <<= 1; return 32 * p; }
    const int k64 = (K + 63) & ~63;
    return k64 <= 512 ? k64 : 0;
}

// ------------------------------------------------------------------ partial reduce + apply
template <bool WSTEP>
__global__ __launch_bounds__(256) void apply_partials_kernel(float *__restrict__ U, const float *__restrict__ P, int nsplit,
                                                             const float *__restrict__ nrm, size_t count, int Mp, int Kp) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float s = P[i];
        for (int sp = 1; sp < nsplit; ++sp) s += P[(size_t)sp * count + i];
        const int k = WSTEP ? (int)(i / (size_t)Mp) : (int)(i % (size_t)Kp);
        U[i] = U[i] * (s / nrm[k]);
    }
}

static inline unsigned ew_grid(size_t n) {
    size_t g = (n + 255) / 256;
    if (g > 256 * 16) g = 256 * 16;
    if (g < 1) g = 1;
    return (unsigned)g;
}

hipError_t launch_apply_partials(float *U, const float *partials, int nsplit, const float *norm, int Mp, int Np, int Kp,
                                 bool wstep, hipStream_t stream) {
    const size_t count = wstep ? (size_t)Mp * Kp : (size_t)Kp * Np;
    if (wstep) hipLaunchKernelGGL(apply_partials_kernel<true>, dim3(ew_grid(count)), dim3(256), 0, stream, U, partials, nsplit, norm, count, Mp, Kp);
    else       hipLaunchKernelGGL(apply_partials_kernel<false>, dim3(ew_grid(count)), dim3(256), 0, stream, U, partials, nsplit, norm, count, Mp, Kp);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void sum_partials_kernel(float *__restrict__ out, const float *__restrict__ P, int nsplit, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        float s = P[i];
        for (int sp = 1; sp < nsplit; ++sp) s += P[(size_t)sp * count + i];
        out[i] = s;
    }
}
hipError_t launch_sum_partials(float *psum, const float *partials, int nsplit, size_t count, hipStream_t stream) {
    hipLaunchKernelGGL(sum_partials_kernel, dim3(ew_grid(count)), dim3(256), 0, stream, psum, partials, nsplit, count);
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

// =====================================================================================
// Convergence check: KL(X || WH), sum|X - WH|, sum|X|  (reduce1d_div / reduce1d_diff,
// cuda/matrix.cu:505-640) fused behind product 1 so W*H is never materialised.
// =====================================================================================
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
            if (x > 0.f) {   // padding is exactly 0; real inputs are >= EPS (cuda/nmf.cu:211)
                fkl += x * (logf(x) - logf(y)) - x + y;   // cuda/matrix.cu:592
                fd += fabsf(x - y);                       // cuda/matrix.cu:517
                fx += fabsf(x);                           // cuda/matrix.cu:518
            }
        }
        kl += (double)fkl; dabs += (double)fd; xabs += (double)fx;
        if (more) load_x<false>(xr, X, Mp, (ch + 1) * 32, q0, c, h);
        if (more) stage_store<KT, false>(st, vn, tid);
        __syncthreads();
    }
    if (!active) { kl = 0.0; dabs = 0.0; xabs = 0.0; }
    block_reduce3(kl, dabs, xabs, part + 3 * (size_t)blockIdx.x, tid);
}

int check_num_groups(int Np, int Kp) { return (Np + fused_cols_per_group(Kp) - 1) / fused_cols_per_group(Kp); }

template <int KT>
static hipError_t launch_check_kt(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream) {
    // the production half-step kernel in CHECK mode (product 1 + KL terms); NMF_FUSED_VARIANT=1 keeps the first-generation check_kernel
    if (fused_variant() != 1 && (size_t)Kp * (size_t)Mp < ((size_t)1 << 31)) {
        FusedArgs a;
        a.W = W; a.H = H; a.X = X; a.U_out = nullptr; a.partials = nullptr; a.norm = nullptr;
        a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.nsplit = 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
        const size_t lds3 = (size_t)2 * KT * 32 * kLdv * sizeof(float) + 4 * kXtFloats * sizeof(float);
        {
            hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_v3<KT, false, false, 0, false, true>, lds3);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL((fused_step_kernel_v3<KT, false, false, 0, false, true>), dim3(check_num_groups(Np, Kp)), dim3(256), lds3, stream, a, part);
        return hipGetLastError();
    }
    const size_t lds = (size_t)2 * KT * 32 * kLdv * sizeof(float);
    {
        hipError_t e = ensure_dynamic_lds((const void *)check_kernel<KT>, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((check_kernel<KT>), dim3(check_num_groups(Np, Kp)), dim3(256), lds, stream, W, H, X, Mp, Np, Kp, part);
    return hipGetLastError();
}

hipError_t launch_check(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream) {
    if ((Mp | Np | Kp) & 31) return hipErrorInvalidValue;
    if (use_k16(Kp)) {
        switch (Kp / 64) {
            case 1: return launch_check_k16<1, 2>(W, H, X, Mp, Np, Kp, part, stream);
            case 2: return launch_check_k16<2, 2>(W, H, X, Mp, Np, Kp, part, stream);
            case 4: return launch_check_k16<4, 2>(W, H, X, Mp, Np, Kp, part, stream);
            case 5: return launch_check_k16<5, 1>(W, H, X, Mp, Np, Kp, part, stream);
            case 6: return launch_check_k16<6, 1>(W, H, X, Mp, Np, Kp, part, stream);
            case 7: return launch_check_k16<7, 1>(W, H, X, Mp, Np, Kp, part, stream);
            case 8: return launch_check_k16<8, 1>(W, H, X, Mp, Np, Kp, part, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (Kp / 32) {
        case 1: return launch_check_kt<1>(W, H, X, Mp, Np, Kp, part, stream);
        case 2: return launch_check_kt<2>(W, H, X, Mp, Np, Kp, part, stream);
        case 4: return launch_check_kt<4>(W, H, X, Mp, Np, Kp, part, stream);
        case 8: return launch_check_kt<8>(W, H, X, Mp, Np, Kp, part, stream);
        default: return hipErrorInvalidValue;
    }
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
__global__ __launch_bounds__(256) void col_sums_kernel(const float *__restrict__ A, int rows, long ld, float *__restrict__ out, int clamp) {
    __shared__ float red[4];
    const float *__restrict__ a = A + (size_t)blockIdx.x * ld;
    float s = 0.f;
    for (int i = threadIdx.x; i < rows; i += 256) s += a[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        out[blockIdx.x] = clamp ? clamp_eps(tot) : tot;
    }
}
hipError_t launch_col_sums(const float *A, int rows, int cols, long ld, float *out, bool clamp, hipStream_t stream) {
    hipLaunchKernelGGL(col_sums_kernel, dim3(cols), dim3(256), 0, stream, A, rows, ld, out, clamp ? 1 : 0);
    return hipGetLastError();
}

constexpr int kRowSumCols = 64;   // columns per workgroup in level 1 (>= 1024 workgroups at N = 65536)
int row_sum_blocks(int cols) { return (cols + kRowSumCols - 1) / kRowSumCols; }

// level 1: part[b*rows + k] = sum over this block's columns of A[k + col*ld]; fixed order per (b, k)
__global__ __launch_bounds__(256) void row_sums_l1_kernel(const float *__restrict__ A, int rows, int cols, long ld, float *__restrict__ part) {
    __shared__ float red[256];
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
hipError_t launch_row_sums(const float *A, int rows, int cols, long ld, float *part, float *out, bool clamp, hipStream_t stream) {
    const int nblk = row_sum_blocks(cols);
    hipLaunchKernelGGL(row_sums_l1_kernel, dim3(nblk), dim3(256), 0, stream, A, rows, cols, ld, part);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(row_sums_l2_kernel, dim3((rows + 31) / 32), dim3(1024), 0, stream, part, rows, nblk, out, clamp ? 1 : 0);
    return hipGetLastError();
}

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
