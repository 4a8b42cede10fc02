// nmf_pair16_impl.h -- the fused half-step of update_div (cuda/nmf.cu:118-176) for 512 < K <= 1024, K computed at the
// reference's granularity of 32 (PAD_MULT, cuda/matrix.cuh:7; cuda/matrix.cu:88-95): kernel and launch templates.  The
// instantiations live in nmf_pair16_inst.hip (compiled in four groups side by side), the dispatch in nmf_pair16.hip.
//
// At K = 1024 the K x 16 accumulator of a wave's 16 owned columns (256 registers per lane) and the K operands of product 1
// (256 more) no longer fit one wave.  Here TWO WAVES SHARE 16 OWNED COLUMNS AND SPLIT K: wave h of a pair holds the B operands
// and the accumulator rows of k in [h K/2, (h + 1) K/2).  Per 16-row chunk of the streamed factor
//   1. each wave forms its half of S = V_chunk * U over its k range (two interleaved MFMA chains),
//   2. the halves meet in LDS (1 KiB per wave), both waves add the two halves (a + b = b + a bit for bit) and take the quotient Z = X ./ max(S, EPS),
//   3. each wave accumulates Acc[k range] += V_chunk[:, k range]' * Z -- no MFMA is issued twice, no accumulator is shared.
// Same v_mfma_f32_16x16x4_f32 lane maps as nmf_fused16_impl.h (lane l: j = l & 15, kq = l >> 4; A = A[row j][k kq],
// B = B[k kq][col j], result register r = D[4 kq + r][j]); chunks are 16 rows so that the K x 16 LDS image (row stride 17)
// can be double-buffered: 2 x 68 KiB at K = 1024.  The X tile (16 x 16) is loaded straight into the accumulator layout.
// CHECK = true: the KL / rel-L1 check for this K range (product 1 + exchange, terms summed by the h = 0 wave of each pair).
//
// Round 5: the template parameter is KTH = 16 x 16 accumulator tiles per wave, K = 32 KTH for KTH = 19 ... 32 (K = 608, 640, ...,
// 1024; K <= 576 stays on the 64-column kernel) -- until then NBH = whole 64-blocks per wave, K = 640 / 768 / 896 / 1024 only, and K = 520 computed on 640.  A wave's half
// KH = 16 KTH of K is walked by product 1 in whole blocks of 64 (per-lane runs of 16) and, where KH % 64 != 0, one remainder block in
// runs of R = KH % 64 / 4 per lane group (k16_kconst / k16_rem_lane, nmf_device.h: the map of the 64-column kernel).  In HBM and in
// LDS the factors are KP = K rounded up to 64 wide (the staging moves whole 64-column pieces; rows K .. KP - 1 are zero padding no
// MFMA reads), so where K % 64 == 32 the image holds two ds_writes per thread more than product 2 has slots: they follow the loop.
#pragma once
#include "nmf_device.h"

namespace nmf {

constexpr int kLdp = 17;   // LDS row stride of the 16-wide image: product-1 reads hit banks 16 kq + j (conflict-free)

// four quotients of one lane, correctly rounded; range-guarded short sequence as quotient8 (nmf_device.h)
template <int DIV>
__device__ __forceinline__ void quotient4(const float (&x)[4], const f32x4 &s, float (&z)[4], bool in_range) {
    if (DIV == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = quotient<1>(x[r], clamp_eps(s[r]));
        return;
    }
    unsigned m = __float_as_uint(s[0]);
#pragma unroll
    for (int r = 1; r < 4; ++r) { const unsigned b = __float_as_uint(s[r]); m = b > m ? b : m; }
    const bool fast = in_range && __builtin_amdgcn_ballot_w64(m > __float_as_uint(kDivSafeMax)) == 0;
    if (fast) {
        // quotient<1>, stage by stage on two packed pairs (v_pk_fma_f32 / v_pk_mul_f32: half the instructions of the scalar form), as quotient8
        const float eps = kEps;
        float y[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) asm("v_max_f32 %0, %1, %2" : "=v"(y[r]) : "v"(s[r]), "v"(eps));
        f32x2 yy[2], xx[2], rc[2], q[2], e[2];
        const f32x2 one = {1.0f, 1.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i) { yy[i] = f32x2{y[2 * i], y[2 * i + 1]}; xx[i] = f32x2{x[2 * i], x[2 * i + 1]}; }
#pragma unroll
        for (int i = 0; i < 2; ++i) rc[i] = f32x2{__builtin_amdgcn_rcpf(yy[i].x), __builtin_amdgcn_rcpf(yy[i].y)};
#pragma unroll
        for (int i = 0; i < 2; ++i) e[i] = __builtin_elementwise_fma(-yy[i], rc[i], one);
#pragma unroll
        for (int i = 0; i < 2; ++i) rc[i] = __builtin_elementwise_fma(e[i], rc[i], rc[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) q[i] = xx[i] * rc[i];
#pragma unroll
        for (int i = 0; i < 2; ++i) e[i] = __builtin_elementwise_fma(-yy[i], q[i], xx[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) q[i] = __builtin_elementwise_fma(e[i], rc[i], q[i]);
#pragma unroll
        for (int i = 0; i < 2; ++i) { z[2 * i] = q[i].x; z[2 * i + 1] = q[i].y; }
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) z[r] = x[r] / clamp_eps(s[r]);
    }
}

// KTH = 16-row accumulator tiles per wave: K = 32 KTH
template <int KTH, bool WSTEP, bool PARTIAL, int DIV, bool CHECK = false>
__global__ __launch_bounds__(256, 1) void fused_step_kernel_pair(FusedArgs a, double *__restrict__ chk_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int KH = 16 * KTH, K = 2 * KH, KP = (K + 63) & ~63;
    constexpr int VBUF = KP * kLdp;
    constexpr int N1 = 4 * KTH;      // product-1 steps of this wave (4 k each)
    constexpr int NF = 16 * (KTH / 4);   // ... of which in whole 64-blocks of k (run map)
    constexpr int RR = N1 - NF;      // ... and in the remainder block: 0, 4, 8 or 12, runs of RR per lane group
    constexpr int NT = KTH;          // 16 x 16 accumulator tiles of this wave
    constexpr int NST = KP / 64;     // staged 16-B pieces per thread per chunk (KP * 16 floats / 256 threads / 4)
    constexpr int D = kRing;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    const int pair = wave >> 1, h = wave & 1;
    const int P = WSTEP ? a.Np : a.Mp;
    const int nsplit = a.nsplit;
    const bool x_in_range = a.x_in_range != 0;
    const int split = blockIdx.x % nsplit;
    const int qblk = blockIdx.x / nsplit;
    const int q0 = (qblk * 2 + pair) * 16;          // Q is a multiple of 32: always inside
    const float *__restrict__ V = WSTEP ? a.H : a.W;
    const float *__restrict__ U = WSTEP ? a.W : a.H;
    const long ldv = WSTEP ? a.Kp : a.Mp, ldu = WSTEP ? a.Mp : a.Kp, ldx = a.Mp;
    const int nchunks = P / 16;
    const int cps = (nchunks + nsplit - 1) / nsplit;
    const int c_begin = split * cps;
    const int c_end = (c_begin + cps < nchunks) ? (c_begin + cps) : nchunks;
    const int kbase = h * KH;

    // B operands of product 1: ub[s] = U(kbase + k(s, kq), q0 + j), k(s, kq) = 64 (s >> 4) + 16 kq + (s & 15) in the whole blocks,
    // 4 NF + RR pi(kq) + (s - NF) in the remainder block (pi = (0, 2, 1, 3))
    const int krem = RR ? k16_rem_lane<RR ? RR : 4, false>(kq) : 0;
    float ub[N1];
    if (!WSTEP) {
        const float *__restrict__ col = U + (size_t)kbase + (size_t)(q0 + j) * ldu;
#pragma unroll
        for (int sb = 0; sb < NF / 16; ++sb)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 16 * kq + 64 * sb + 4 * e4);
                ub[16 * sb + 4 * e4] = v[0]; ub[16 * sb + 4 * e4 + 1] = v[1]; ub[16 * sb + 4 * e4 + 2] = v[2]; ub[16 * sb + 4 * e4 + 3] = v[3];
            }
#pragma unroll
        for (int e4 = 0; e4 < RR / 4; ++e4) {
            const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 4 * NF + krem + 4 * e4);
            ub[NF + 4 * e4] = v[0]; ub[NF + 4 * e4 + 1] = v[1]; ub[NF + 4 * e4 + 2] = v[2]; ub[NF + 4 * e4 + 3] = v[3];
        }
    } else {
#pragma unroll
        for (int s = 0; s < N1; ++s)
            ub[s] = U[(size_t)(q0 + j) + (size_t)(kbase + k16_kconst<NF, false>(s) + (k16_in_rem<NF>(s) ? krem : 16 * kq)) * ldu];
    }

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    double kl = 0.0, dabs = 0.0;
    f32x4 *xch = reinterpret_cast<f32x4 *>(smem + 2 * VBUF);   // [wave][lane]: the halves of S

    if (c_begin < c_end) {
        // staging: H-step piece = rows 4 i4 .. + 3 of column k = kk + 64 q of W (i4 = tid & 3, kk = tid >> 2);
        //          W-step piece = rows 4 (k4 + 16 q) .. + 3 of column p = tid >> 4 of H (k4 = tid & 15)
        const unsigned voff0 = WSTEP ? 4u * ((unsigned)(4 * (tid & 15)) + (unsigned)(tid >> 4) * (unsigned)ldv)
                                     : 4u * ((unsigned)(4 * (tid & 3)) + (unsigned)(tid >> 2) * (unsigned)ldv);
        const unsigned vstep = 4u * (WSTEP ? 64u : 64u * (unsigned)ldv);
        const size_t vchunk = 4 * (WSTEP ? (size_t)16 * (size_t)ldv : (size_t)16);
        const int st_off = WSTEP ? (4 * (tid & 15)) * kLdp + (tid >> 4) : (tid >> 2) * kLdp + 4 * (tid & 3);   // + 64 q * kLdp, + cc (H) / + cc * kLdp (W)
        // X tile (16 p x 16 q) straight into the accumulator layout: lane needs X(p0 + 4 kq + r, q0 + j), r = 0..3
        const unsigned xoff0 = WSTEP ? 4u * ((unsigned)j + (unsigned)(4 * kq) * (unsigned)ldx) : 4u * ((unsigned)(4 * kq) + (unsigned)j * (unsigned)ldx);
        const char *__restrict__ xbase = reinterpret_cast<const char *>(WSTEP ? a.X + (size_t)q0 : a.X + (size_t)q0 * (size_t)ldx);
        const size_t xchunk = 4 * (WSTEP ? (size_t)16 * (size_t)ldx : (size_t)16);
        const int p1_off = (kbase + 16 * kq) * kLdp + j;     // + (64 (s >> 4) + (s & 15)) * kLdp
        const int p1r_off = (kbase + krem) * kLdp + j;       // remainder block: + (4 NF + s - NF) * kLdp
        const int p2_off = (kbase + j) * kLdp + 4 * kq;      // + 16 t * kLdp + r

        f32x4 st[NST];
        float xn[4], xr[4];
        unsigned vo = voff0, xo = xoff0;
        const char *__restrict__ vcur = reinterpret_cast<const char *>(V);
        const char *__restrict__ xcur = xbase;
        auto set_chunk = [&](int ch) {
            vo = voff0; xo = xoff0;
            asm volatile("" : "+v"(vo), "+v"(xo));
            vcur = reinterpret_cast<const char *>(V) + (size_t)ch * vchunk;
            xcur = xbase + (size_t)ch * xchunk;
        };
        auto stage_load_one = [&](int q) {
            global_bytes base = (global_bytes)(vcur + (size_t)q * (size_t)vstep);
            asm volatile("" : "+s"(base));
            st[q] = *(const __attribute__((address_space(1))) f32x4 *)(base + vo);
        };
        auto x_load = [&]() {
            if (!WSTEP) {
                global_bytes base = (global_bytes)xcur;
                asm volatile("" : "+s"(base));
                const f32x4 v = *(const __attribute__((address_space(1))) f32x4 *)(base + xo);
                xn[0] = v[0]; xn[1] = v[1]; xn[2] = v[2]; xn[3] = v[3];
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    global_bytes base = (global_bytes)(xcur + (size_t)r * 4 * (size_t)ldx);
                    asm volatile("" : "+s"(base));
                    xn[r] = *(const __attribute__((address_space(1))) float *)(base + xo);
                }
            }
        };
        auto stage_store_one = [&](float *__restrict__ vl, int w) {
            const int q = w / 4, cc = w % 4;
            if (!WSTEP) vl[st_off + 64 * q * kLdp + cc] = st[q][cc];
            else        vl[st_off + (64 * q + cc) * kLdp] = st[q][cc];
        };

        set_chunk(c_begin);
#pragma unroll
        for (int q = 0; q < NST; ++q) stage_load_one(q);
        x_load();
#pragma unroll
        for (int w = 0; w < 4 * NST; ++w) stage_store_one(smem, w);
#pragma unroll
        for (int r = 0; r < 4; ++r) xr[r] = xn[r];
        __syncthreads();
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int par = (ch - c_begin) & 1;
            const float *__restrict__ vb = smem + par * VBUF;
            float *__restrict__ vn = smem + (par ^ 1) * VBUF;
            set_chunk((ch + 1 < c_end) ? ch + 1 : ch);
            // ---- product 1 over this wave's k range: two interleaved chains (even / odd steps), summed afterwards.  Builtin MFMAs, not
            // inline asm: this kernel keeps part of its registers in AGPRs, and a copy the compiler places right in front of an
            // asm MFMA is a hazard it cannot see (nmf_split16.hip, K = 256); the builtin costs nothing here (137 TFLOP/s either way)
            const lds_float *b1 = (const lds_float *)vb + p1_off;
            const lds_float *b1r = (const lds_float *)vb + p1r_off;
            auto p1_read = [&](int e) { return lds_ld((k16_in_rem<NF>(e) ? b1r : b1) + k16_kconst<NF, false>(e) * kLdp); };
            float ar[D];
#pragma unroll
            for (int e = 0; e < D; ++e) ar[e] = p1_read(e);
            f32x4 s0, s1;
            constexpr int NLOAD = NST + 1;                 // the X tile first (used first), then the pieces of the next image
            constexpr int G = CHECK ? (N1 / 2) / (NLOAD + 1) : N1 / (NLOAD + 1);
#pragma unroll
            for (int e = 0; e < N1; ++e) {
                if (e == 0)      s0 = NMF_MFMA16(ar[0], ub[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                else if (e == 1) s1 = NMF_MFMA16(ar[1], ub[1], (f32x4{0.f, 0.f, 0.f, 0.f}));
                else if (e & 1)  s1 = NMF_MFMA16(ar[e % D], ub[e], s1);
                else             s0 = NMF_MFMA16(ar[e % D], ub[e], s0);
                if (e + D < N1) ar[e % D] = p1_read(e + D);
                if (e >= G && e % G == 0 && e / G - 1 < NLOAD) {
                    const int l = e / G - 1;
                    if (l == 0) x_load(); else stage_load_one(l - 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (CHECK && e >= N1 / 2) {   // no product 2 to hide the image behind: one ds_write per MFMA of the second half (N1 / 2 = K / 16 of the KP / 16)
                    stage_store_one(vn, e - N1 / 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (CHECK) {
#pragma unroll
                for (int w = N1 / 2; w < 4 * NST; ++w) stage_store_one(vn, w);   // K % 64 == 32: the zero padding of the last 64-column piece
            }
            // (no explicit wait states here: every MFMA of this kernel is a builtin, so the compiler's hazard recognizer pads the read of s0 / s1
            //  itself.  Until round 5 an `s_nop 15; s_nop 3` statement stood here, a leftover of the inline-asm chain this kernel started with;
            //  its "+v" operands also pinned both accumulators and the schedule around them: 1-2 % of the half-step, profiles/r05_pair_ablation.log)
            // ---- the two halves of S meet: both waves of the pair add them in the same order
            // (each wave reads its partner's half only: a + b and b + a are the same bits, so both waves of the pair still hold the same S)
            const f32x4 mine = s0 + s1;
            xch[wave * 64 + lane] = mine;
            __syncthreads();
            const f32x4 s = mine + xch[(wave ^ 1) * 64 + lane];
            if (CHECK) {
                if (h == 0) {
                    float fkl = 0.f, fd = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = xr[r], y = clamp_eps(s[r]);
                        fkl = __builtin_fmaf(x, log2_hw(y), fkl); fd += fabsf(x - y);   // see fused_step_kernel_k16<CHECK>
                    }
                    kl += (double)fkl; dabs += (double)fd;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) xr[r] = xn[r];
                __syncthreads();
                continue;
            }
            // ---- first operands of product 2, then the quotient
            const lds_float *b2 = (const lds_float *)vb + p2_off;
            constexpr int E2 = 4 * NT;   // order: r outer, tile t inner
            float a2[D];
#pragma unroll
            for (int e = 0; e < D; ++e) a2[e] = lds_ld(b2 + 16 * (e % NT) * kLdp + (e / NT));
            float z[4];
            __builtin_amdgcn_sched_barrier(0);
            quotient4<DIV>(xr, s, z, x_in_range);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) xr[r] = xn[r];
            // ---- product 2: NT independent accumulators; the next image goes to LDS one ds_write every second MFMA
#pragma unroll
            for (int g = 0; g < 4; ++g) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int e = g * NT + t;
                    acc[t] = NMF_MFMA16(a2[e % D], z[g], acc[t]);
                    if (e + D < E2) {
                        const int en = e + D;
                        a2[e % D] = lds_ld(b2 + 16 * (en % NT) * kLdp + (en / NT));
                    }
                    if (e % 2 == 0 && e / 2 < 4 * NST) {
                        stage_store_one(vn, e / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
#pragma unroll
            for (int w = E2 / 2; w < 4 * NST; ++w) stage_store_one(vn, w);   // K % 64 == 32: two more pieces than product 2 has slots
            __syncthreads();
        }
    }
    if (CHECK) {
        block_reduce3(kl, dabs, 0.0, chk_part + 3 * (size_t)blockIdx.x, tid);
        return;
    }
    // epilogue: lane holds Acc(k = kbase + 16 t + 4 kq + r, q0 + j)
    if (PARTIAL) {
        const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
        float *__restrict__ out = a.partials + (size_t)split * slab;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = kbase + 16 * t + 4 * kq;
            if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)k + (size_t)(q0 + j) * ldu) = acc[t];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(k + r) * ldu] = acc[t][r];
            }
        }
        // rows K .. KP - 1 of a slab are zero padding, and the slab buffer is shared between the half-steps: written as zeros (by the upper wave)
        if (KP > K && h == 1) {
#pragma unroll
            for (int t = 0; t < (KP - K) / 16; ++t) {
                const int k = K + 16 * t + 4 * kq;
                if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)k + (size_t)(q0 + j) * ldu) = f32x4{0.f, 0.f, 0.f, 0.f};
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(k + r) * ldu] = 0.f;
                }
            }
        }
    } else {
        float *__restrict__ Uo = a.U_out;
        const float *__restrict__ nrm = a.norm;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = kbase + 16 * t + 4 * kq;
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

inline size_t pair_lds_bytes(int Kp) { return ((size_t)2 * Kp * kLdp + 4 * 64 * 4) * sizeof(float); }

template <int KTH>
hipError_t launch_pair_kth(const FusedArgs &a, bool wstep, hipStream_t stream) {
    const int Q = wstep ? a.Mp : a.Np;
    const dim3 grid((unsigned)((Q / 32) * a.nsplit)), block(256);
    const size_t lds = pair_lds_bytes(a.Kp);
    const bool partial = a.partial != 0;
    // (no DIV = 1 instantiations since round 5: nmf_fused16_impl.h, launch_fused_k16)
#define NMF_LAUNCH_P16(...)                                                                               \
    do {                                                                                                  \
        hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                                \
        if (e != hipSuccess) return e;                                                                    \
        note_kernel((const void *)__VA_ARGS__, stream);                                                   \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a, (double *)nullptr);                \
    } while (0)
    if (!wstep && !partial) NMF_LAUNCH_P16(fused_step_kernel_pair<KTH, false, false, 0>);
    else if (!wstep && partial) NMF_LAUNCH_P16(fused_step_kernel_pair<KTH, false, true, 0>);
    else if (wstep && !partial) NMF_LAUNCH_P16(fused_step_kernel_pair<KTH, true, false, 0>);
    else NMF_LAUNCH_P16(fused_step_kernel_pair<KTH, true, true, 0>);
#undef NMF_LAUNCH_P16
    return hipGetLastError();
}

template <int KTH>
hipError_t launch_check_kth(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream) {
    FusedArgs a;
    a.W = W; a.H = H; a.X = X; a.U_out = nullptr; a.partials = nullptr; a.norm = nullptr;
    a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.nsplit = 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
    const size_t lds = pair_lds_bytes(Kp);
    hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_pair<KTH, false, false, 0, true>, lds);
    if (e != hipSuccess) return e;
    note_kernel((const void *)fused_step_kernel_pair<KTH, false, false, 0, true>, stream);
    hipLaunchKernelGGL((fused_step_kernel_pair<KTH, false, false, 0, true>), dim3(Np / 32), dim3(256), lds, stream, a, part);
    return hipGetLastError();
}

// KTH = K / 32 for every K the family serves (608 ... 1024: K <= 576 stays on the 64-column kernel), in four groups of about equal compile time (the groups build side by side)
#define NMF_P16_GROUP0(X) X(32) X(21) X(20)
#define NMF_P16_GROUP1(X) X(31) X(28) X(23)
#define NMF_P16_GROUP2(X) X(30) X(27) X(24) X(19)
#define NMF_P16_GROUP3(X) X(29) X(26) X(25) X(22)
#define NMF_P16_ALL(X) NMF_P16_GROUP0(X) NMF_P16_GROUP1(X) NMF_P16_GROUP2(X) NMF_P16_GROUP3(X)

}  // namespace nmf
