// nmf_fused16_impl.h -- the production fused half-step of update_div (cuda/nmf.cu:118-176) for K <= 576:
// 16 owned columns per wave on v_mfma_f32_16x16x4_f32, and the KL check built from the same code.
// Included by nmf_fused16.hip (dispatch) and nmf_fused16_inst.hip (the instantiations, compiled in groups).
#pragma once
#include "nmf_device.h"

namespace nmf {

// =====================================================================================
// Same algorithm as v3 (nmf_fused32.hip) on v_mfma_f32_16x16x4_f32 with 16 owned columns per wave, so that the K x 16
// accumulator (4 KT registers) and the K B-operands of product 1 (4 KT registers) fit the register file.
// The template parameter is KT = K / 16, the number of 16-row accumulator tiles: the reference pads K to a multiple of 32 and
// nothing coarser (cuda/matrix.cuh:7, cuda/matrix.cu:88-95), and NMF ranks are rarely powers of two, so the kernel is
// instantiated for every multiple of 16 up to 512.  The factors in HBM and the LDS image
// are padded to KS = 32 ceil(KT / 2) columns (= FusedArgs::Kp; zeros beyond K); the MFMAs cover K = 16 KT (= FusedArgs::Kc).
// Lane maps of the 16x16x4 form (lane l: j = l & 15, kq = l >> 4):
//     A operand = A[row j][k kq],  B operand = B[k kq][col j],  result reg r = D[4 kq + r][j]
// so register r of a finished 16x16 tile is the B operand of a step whose four k indices are the tile rows 4 kq + r: the
// quotient again feeds product 2 straight from the accumulator registers.
// Product 1 runs two interleaved chains (the two 16-row halves of the 32-row chunk): the 16x16x4 MFMA issues every 32 cycles
// but needs 40 between dependent ones.
// k index of product-1 step s in lane group kq (k16_kidx): whole blocks of 64 first, 64 (s >> 4) + 16 kq + (s & 15) --
// per-lane contiguous runs of 16 (16-B loads of the owned factor) and, with 33-float LDS rows, 32 distinct banks per
// half-wave -- then, where K is not a multiple of 64, one remainder block of 4 R (R = 4, 8, 12) in runs of R:
// 64 (K / 64) + R pi(kq) + s', pi = (0, 2, 1, 3), so that the two lane groups of a half-wave sit 2 R apart (conflict-free
// at R = 8; 2-way on half of the lanes at R = 4, 12: ds_read issue is hidden behind the MFMAs either way).
// CHECK = true turns the kernel into the KL / rel-L1 check (product 1 only), see check_kernel.
// GEMM = true (H-step orientation only) keeps product 1 alone and stores it: C = W * H for K <= 512, the reference's
// matrix_multiply (cuda/matrix.cu:97-105) on the W*H shape at the rate of the fused loop (a.U_out = C, ld = Mp; a.X unused).
// =====================================================================================
constexpr int kXt16Floats = 32 * 20;   // per-wave X patch: H-step 16 x 36, W-step 32 x 20 floats

template <int KT, bool WSTEP, bool PARTIAL, int DIV, bool CHECK = false, int OCC = 1, bool GEMM = false, int TRIM = 0>
__global__ __launch_bounds__(256, OCC) void fused_step_kernel_k16(FusedArgs a, double *__restrict__ chk_part) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 16 * KT;       // what the MFMAs cover (FusedArgs::Kc)
    constexpr int NST = (KT + 1) / 2;   // staged 16-B pieces per thread per chunk: 32 columns of the streamed factor each
    constexpr int KS = 32 * NST;     // staged columns = FusedArgs::Kp (zero padding beyond K)
    constexpr int VBUF = KS * kLdv;
    constexpr int N1 = 4 * KT;       // product-1 steps per 16-row tile
    constexpr int NT = KT;           // 16 x 16 accumulator tiles
    // The remainder block of product 1: the steps beyond the whole blocks of 64 -- or, in a TRIM variant of a K that is a multiple of
    // 64 (K = 50 on the K = 64 kernel), the last whole block (16 steps), so that those variants have something to trim.  The TRIM = 0
    // kernels of K = 64 / 128 / 192 / 256 keep the run map throughout: interleaving their last block cost K = 64 2.5 % when it was
    // tried on the one kernel that served both (profiles/r04_last_block_interleave.log).
    constexpr bool LASTIL = TRIM > 0 && KT % 4 == 0;
    constexpr int NF = LASTIL ? 16 * (KT / 4 - 1) : 16 * (KT / 4);   // product-1 steps in the whole 64-blocks of k that use the run map
    constexpr int RR = N1 - NF;         // product-1 steps in the remainder block (0, 4, 8, 12; 16 under LASTIL)
    constexpr int D = kRing;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    // the last p1_trim (0 .. 3) steps of the remainder block cover zero padding only (K_true <= K - 4 p1_trim): skipped
    // TRIM = 2, 3 (K <= 256; chosen by the launcher when the caller's K leaves the last two / three steps of product 1 on
    // zero padding, e.g. K = 100 on the K = 112 kernel: 25 steps instead of 28; K = 200 on the K = 208 kernel: 50 of 52): the remainder block is interleaved over the lane groups so
    // that its last steps cover the top k indices, and the chain simply ends that many steps early -- compile-time variants, because a
    // run-time switch (one uniform branch in front of the chain's tail, which forces compiler-placed MFMAs: the copies at the join sit
    // next to MFMAs whose hazards it must see) cost the untrimmed kernels 0.2 .. 1 % and the split kernel 9 .. 12 %
    // (profiles/r04_trim_ab.log, r04_small_levers.log).
    static_assert(TRIM == 0 || ((TRIM == 2 || TRIM == 3) && RR >= 4 && KT <= 16 && !CHECK && !GEMM), "the trimmed chains exist for the K <= 256 half-steps");
    static_assert(KT > 1 || TRIM < 3, "K = 16 has four steps: its chain may end two early, not three");
    constexpr bool IL = TRIM > 0;
    constexpr int N1R = N1 - TRIM;      // product-1 steps that are issued
    const int rl = k16_rem_lane<RR, IL>(kq);   // lane part of the k index in the remainder block
    const int P = WSTEP ? a.Np : a.Mp;
    const int Q = WSTEP ? a.Mp : a.Np;
    const int nsplit = a.nsplit;
    const bool x_in_range = a.x_in_range != 0;
    const int split = blockIdx.x % nsplit;
    const int qblk = blockIdx.x / nsplit;
    int q0 = (qblk * 4 + wave) * 16;
    const bool active = q0 < Q;
    if (!active) q0 = Q - 16;
    const size_t pair = GEMM ? 0 : (size_t)blockIdx.y;   // pair of a batched solver: one launch over all of them (FusedArgs::strideW, batch)
    if (!CHECK && !GEMM && a.active != nullptr && a.active[pair] == 0) return;   // this pair has converged (uniform: ahead of every barrier)
    const float *__restrict__ V = (WSTEP ? a.H : a.W) + pair * (WSTEP ? a.strideH : a.strideW);
    const float *__restrict__ U = (WSTEP ? a.W : a.H) + pair * (WSTEP ? a.strideW : a.strideH);
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
        for (int sb = 0; sb < NF / 16; ++sb)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 64 * sb + 4 * e4);
                ub[16 * sb + 4 * e4] = v[0]; ub[16 * sb + 4 * e4 + 1] = v[1]; ub[16 * sb + 4 * e4 + 2] = v[2]; ub[16 * sb + 4 * e4 + 3] = v[3];
            }
        if (RR > 0) {   // the remainder block: one dword per step (interleaved) or a run of RR (a multiple of 4: 16-B loads)
            const float *__restrict__ colr = U + (size_t)(4 * NF + rl) + (size_t)(q0 + j) * ldu;
            if (IL) {
#pragma unroll
                for (int sr = 0; sr < RR; ++sr) ub[NF + sr] = colr[4 * sr];
            } else {
#pragma unroll
                for (int e4 = 0; e4 < RR / 4; ++e4) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(colr + 4 * e4);
                    ub[NF + 4 * e4] = v[0]; ub[NF + 4 * e4 + 1] = v[1]; ub[NF + 4 * e4 + 2] = v[2]; ub[NF + 4 * e4 + 3] = v[3];
                }
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < N1; ++s) ub[s] = U[(size_t)(q0 + j) + (size_t)(k16_kconst<NF, IL>(s) + (k16_in_rem<NF>(s) ? rl : 16 * kq)) * ldu];
    }

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    double kl = 0.0, dabs = 0.0, xabs = 0.0;
    // FusedArgs::vsum_part: wave w of q-block b sums row 4 b + w of H over this split's columns
    const bool vsum_on = WSTEP && PARTIAL && a.vsum_part != nullptr;
    // lanes 0 .. 31 sum row 4 b + w over the chunk's 32 columns, lanes 32 .. 63 the row one pass further on (+ 4 x the q-blocks of a split):
    // a split's workgroups cover 8 Mp / 64 rows at no extra instruction.  All KS rows of the image: the padding rows sum to zero
    const int vrow_raw = qblk * 4 + wave + (lane >> 5) * (4 * (int)(gridDim.x / (unsigned)nsplit)), vrow = vrow_raw < KS ? vrow_raw : KS - 1;
    float vs_acc = 0.f;

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
        const int p1_off = 16 * kq * kLdv + j;     // + k16_kconst(s) * kLdv + 16 T
        const int p1r_off = rl * kLdv + j;         // the same for the steps of the remainder block
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
        // An LDS access's immediate offset ends at 64 KiB; an image of K > 496 is longer (K = 576: 74 KiB) and every access beyond would get a
        // v_add_u32 of its own (81 per chunk at K = 576: VALU work the f32 MFMA cannot overlap).  FAR kernels keep second bases 64 KiB further on,
        // made opaque ONCE, outside the chunk loop (inside it an asm statement pins the schedule: tried, slower -- HISTORY.md); the loop adds the
        // image's parity offset to them.
        constexpr int kFar = 16384;   // floats
        constexpr bool FAR = VBUF > kFar;
        typedef __attribute__((address_space(3))) float lds_wfloat;
        auto stage_store_one = [&](float *__restrict__ vl, lds_wfloat *vlf, int w) {
            const int q = w / 4, cc = w % 4;
            const int off = !WSTEP ? 32 * q * kLdv + cc : (32 * q + cc) * kLdv;      // the part of the index the thread does not decide
            if (FAR && off >= kFar) vlf[off - kFar] = st[q][cc];                    // vlf already holds the thread's part
            else if (!WSTEP) { const int k = (tid >> 3) + 32 * q, i4 = tid & 7; vl[k * kLdv + 4 * i4 + cc] = st[q][cc]; }
            else             { const int k4 = q * 8 + (tid & 7), i = (tid >> 3) & 31; vl[(4 * k4 + cc) * kLdv + i] = st[q][cc]; }
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
        if (!GEMM) {
#pragma unroll
            for (int i = 0; i < 2; ++i) x_load_one(i);
        }
#pragma unroll
        for (int w = 0; w < 4 * NST; ++w) stage_store_one(smem, (lds_wfloat *)smem + (!WSTEP ? (tid >> 3) * kLdv + 4 * (tid & 7) : 4 * (tid & 7) * kLdv + ((tid >> 3) & 31)) + kFar, w);
        if (!GEMM) x_relayout();
        __syncthreads();
        const lds_float *far1 = (const lds_float *)smem + p1_off + kFar, *far1r = (const lds_float *)smem + p1r_off + kFar, *far2 = (const lds_float *)smem + p2_off + kFar;
        lds_wfloat *farst = (lds_wfloat *)smem + (!WSTEP ? (tid >> 3) * kLdv + 4 * (tid & 7) : 4 * (tid & 7) * kLdv + ((tid >> 3) & 31)) + kFar;
        if (FAR) asm volatile("" : "+v"(far1), "+v"(far1r), "+v"(far2), "+v"(farst));
        for (int ch = c_begin; ch < c_end; ++ch) {
            const int par = (ch - c_begin) & 1;
            const float *__restrict__ vb = smem + par * VBUF;
            float *__restrict__ vn = smem + (par ^ 1) * VBUF;
            const lds_float *b1f = far1 + par * VBUF, *b1rf = far1r + par * VBUF, *b2f = far2 + par * VBUF;
            lds_wfloat *vnf = farst + (par ^ 1) * VBUF;
            const int chn = (ch + 1 < c_end) ? ch + 1 : ch;
            set_chunk(chn);
            // ---- product 1: two interleaved chains, step index e = 2 s + T
            const lds_float *b1 = (const lds_float *)vb + p1_off;
            const lds_float *b1r = (const lds_float *)vb + p1r_off;
            auto a1_ld = [&](int e) {
                const int off = k16_kconst<NF, IL>(e >> 1) * kLdv + 16 * (e & 1);
                const bool rem = k16_in_rem<NF>(e >> 1);
                return (FAR && off >= kFar) ? lds_ld((rem ? b1rf : b1f) + (off - kFar)) : lds_ld((rem ? b1r : b1) + off);
            };
            // W-step side product: this wave's row of the streamed H chunk, summed per lane (p = lane & 31) over the chunks
            float vs_in = 0.f;
            if (WSTEP && PARTIAL) { if (vsum_on) vs_in = lds_ld((const lds_float *)vb + vrow * kLdv + (lane & 31)); }
            constexpr int E1 = 2 * N1R;
            float ar[D];
#pragma unroll
            for (int e = 0; e < D; ++e) ar[e] = a1_ld(e);
            if (OCC > 1) __builtin_amdgcn_s_setprio(1);
            f32x4 s0, s1;
            constexpr int NLOAD = NST + 2;
            constexpr int G = E1 / (NLOAD + 1);
#pragma unroll
            for (int e = 0; e < E1; ++e) {
                const int s = e >> 1;
                if (KT > 24) {
                    // K > 384 keeps values beyond the accumulator in AGPRs; a copy the compiler makes for an inline-asm MFMA sits right
                    // in front of it and its hazard recogniser cannot see into the asm (the split kernel at K = 256 came out ~1 % wrong
                    // that way).  The builtin is an instruction the compiler knows; it costs nothing here (cfg5 shard: 140.9 TFLOP/s).
                    if (e == 0)      s0 = NMF_MFMA16(ar[0], ub[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    else if (e == 1) s1 = NMF_MFMA16(ar[1], ub[0], (f32x4{0.f, 0.f, 0.f, 0.f}));
                    else if (e & 1)  s1 = NMF_MFMA16(ar[e % D], ub[s], s1);
                    else             s0 = NMF_MFMA16(ar[e % D], ub[s], s0);
                }
                else if (e == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(s0) : "v"(ar[0]), "v"(ub[0]));
                else if (e == 1) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=v"(s1) : "v"(ar[1]), "v"(ub[0]));
                else if (e & 1)  asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(s1) : "v"(ar[e % D]), "v"(ub[s]));
                else             asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(s0) : "v"(ar[e % D]), "v"(ub[s]));
                if (e + D < E1) ar[e % D] = a1_ld(e + D);
                if (GEMM || CHECK) {
                    // no second product to hide the next chunk's LDS image behind: its global loads go into the first half
                    // of this chain, its ds_writes between the MFMAs of the second half (one every other MFMA: the last 8 NST steps,
                    // which is the second half exactly where KT is even)
                    constexpr int NLD = GEMM ? NST : NST + 2;   // the check also fetches the next X tile (first: it is used first)
                    constexpr int GL = (E1 / 2) / (NLD + 1) > 0 ? (E1 / 2) / (NLD + 1) : 1;
                    constexpr int ES = KT == 1 ? 4 : E1 - 8 * NST;   // K = 16: eight steps in all, the loads sit in the first four,
                    constexpr int SS = KT == 1 ? 1 : 2;              //         so a ds_write behind each MFMA of the last four
                    if (e >= GL && e % GL == 0 && e / GL - 1 < NLD) {
                        const int l = e / GL - 1;
                        if (GEMM) stage_load_one(l); else if (l < 2) x_load_one(l); else stage_load_one(l - 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (e >= ES && (e - ES) % SS == 0 && (e - ES) / SS < 4 * NST) {
                        stage_store_one(vn, vnf, (e - ES) / SS);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else if (e >= G && e % G == 0 && e / G - 1 < NLOAD) {
                    const int l = e / G - 1;
                    if (l < 2) x_load_one(l); else stage_load_one(l - 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // An MFMA's result needs 12 wait states (8 passes + 4) before anything but an accumulating MFMA touches it, and the compiler
            // pads nothing behind an asm MFMA.  GEMM / CHECK read it at once: twelve nops.  The half-step first issues the eight LDS
            // reads of product 2's operand ring, which count: six nops behind them (tools/asm_audit.py checks every kernel's code).
            if (GEMM || CHECK) asm volatile("s_nop 11" : "+v"(s0), "+v"(s1));
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            if (GEMM) {   // lane holds S(p0 + 16 T + 4 kq + r, q0 + j): two 16-B stores per chunk, 64 B contiguous per column and half
                if (active) {
                    float *c = a.U_out + (size_t)(q0 + j) * (size_t)a.Mp + (size_t)ch * 32 + 4 * kq;
                    *reinterpret_cast<f32x4 *>(c) = s0;
                    *reinterpret_cast<f32x4 *>(c + 16) = s1;
                }
                __syncthreads();
                continue;
            }
            if (CHECK) {
                float fkl = 0.f, fd = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const float x = xr[r], y = clamp_eps(r < 4 ? s0[r] : s1[r - 4]);
                    fkl = __builtin_fmaf(x, log2_hw(y), fkl); fd += fabsf(x - y);   // the X-only terms of cuda/matrix.cu:592,517-518 are summed once at upload
                }
                kl += (double)fkl; dabs += (double)fd;
                x_relayout();
                __syncthreads();
                continue;
            }
            // ---- first operands of product 2, then the quotient in one VALU block
            const lds_float *b2 = (const lds_float *)vb + p2_off;
            auto a2_ld = [&](int t, int g) {
                const int off = 16 * t * kLdv + 16 * (g >> 2) + (g & 3);
                return (FAR && off >= kFar) ? lds_ld(b2f + (off - kFar)) : lds_ld(b2 + off);
            };
            constexpr int E2 = 8 * NT;   // order: (T, r) outer, tile t inner
            float a2[D];
#pragma unroll
            for (int e = 0; e < D; ++e) a2[e] = a2_ld(e % NT, e / NT);
            float z[8];
            asm volatile("s_nop 5" : "+v"(s0), "+v"(s1));   // D = 8 ds_reads + 6: the wait states of product 1's last MFMAs
            __builtin_amdgcn_sched_barrier(0);
            quotient8<DIV>(xr, s0, s1, z, x_in_range);
            if (WSTEP && PARTIAL) vs_acc += vs_in;
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
                        a2[e % D] = a2_ld(tn, gn);
                    }
                    if (e >= E0 && (e - E0) % 2 == 0 && (e - E0) / 2 < 4 * NST) {
                        stage_store_one(vn, vnf, (e - E0) / 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            __syncthreads();
        }
    }
    if (GEMM) return;
    if (WSTEP && PARTIAL) {
        if (vsum_on) {
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) vs_acc += __shfl_down(vs_acc, off, 32);
            if ((lane & 31) == 0 && vrow_raw < KS) a.vsum_part[(pair * nsplit + (size_t)split) * KS + vrow_raw] = vs_acc;
        }
    }
    if (CHECK) {
        if (!active) { kl = 0.0; dabs = 0.0; xabs = 0.0; }
        block_reduce3(kl, dabs, xabs, chk_part + 3 * ((size_t)blockIdx.x + pair * gridDim.x), tid);
        return;
    }
    if (!active) return;
    // epilogue: lane holds Acc(k = 16 t + 4 kq + r, q0 + j)
    if (PARTIAL) {
        const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
        float *__restrict__ out = a.partials + (pair * nsplit + (size_t)split) * slab;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)(16 * t + 4 * kq) + (size_t)(q0 + j) * ldu) = acc[t];
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(16 * t + 4 * kq + r) * ldu] = acc[t][r];
            }
        }
        if (KT & 1) {   // rows K .. KS - 1 of the slab are zero padding: written, because the slab buffer is shared between the half-steps and
                        // what its consumers multiply into the (zero) padding of U must at least be finite
            if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)(K + 4 * kq) + (size_t)(q0 + j) * ldu) = f32x4{0.f, 0.f, 0.f, 0.f};
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(K + 4 * kq + r) * ldu] = 0.f;
            }
        }
    } else {
        float *__restrict__ Uo = a.U_out + pair * (WSTEP ? a.strideW : a.strideH);
        const float *__restrict__ nrm = a.norm + pair * KS;
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

// the LDS a workgroup of the K = 16 KT kernel needs: two images of the streamed chunk + the four X patches
template <int KT> constexpr size_t k16_lds_bytes() { return ((size_t)2 * 32 * ((KT + 1) / 2) * kLdv + 4 * kXt16Floats) * sizeof(float); }
// two workgroups per CU up to K = 256 (<= 213 registers, <= 78 KiB of LDS), one above
template <int KT> constexpr int k16_occ() { return KT <= 16 ? 2 : 1; }

// The launchers of one KT: defined here, instantiated explicitly in nmf_fused16_inst.hip (one group of KT values per
// compilation), declared `extern template` where they are called from (nmf_fused16.hip).
template <int KT>
hipError_t launch_fused_k16(const FusedArgs &a, bool wstep, hipStream_t stream) {
    constexpr int OCC = k16_occ<KT>();
    const int Q = wstep ? a.Mp : a.Np;
    if (a.batch < 1 || a.batch > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(((Q + 63) / 64) * a.nsplit), (unsigned)a.batch), block(256);
    const size_t lds = k16_lds_bytes<KT>();
    const bool partial = a.partial != 0;
    // (nmf_opts.fast_divide = 1 used to select DIV = 1 instantiations -- the six-instruction quotient without the range guard.  Measured in
    //  round 5 it is 0.1 .. 1.9 % SLOWER than the guarded default on every family (profiles/r05_ab_fast_divide.log: the guard is four
    //  v_max3 and a ballot, and the unguarded variant lost the packed arrangement of the guarded block), so those instantiations --
    //  half of all kernels -- are gone and the option is accepted and ignored.)
#define NMF_LAUNCH_K16(...)                                                                               \
    do {                                                                                                  \
        hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                                \
        if (e != hipSuccess) return e;                                                                    \
        note_kernel((const void *)__VA_ARGS__, stream); \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a, (double *)nullptr);                \
    } while (0)
    if constexpr (KT <= 16) {   // the variants whose product 1 ends two / three steps early (FusedArgs::p1_trim)
#define NMF_LAUNCH_K16_TRIM(T)                                                                                                        \
        {                                                                                                                             \
            if (!wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, false, false, 0, false, OCC, false, T>);                 \
            else if (!wstep && partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, false, true, 0, false, OCC, false, T>);              \
            else if (wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, true, false, 0, false, OCC, false, T>);              \
            else NMF_LAUNCH_K16(fused_step_kernel_k16<KT, true, true, 0, false, OCC, false, T>);                                      \
            return hipGetLastError();                                                                                                 \
        }
        if constexpr (KT > 1) { if (a.p1_trim == 3) NMF_LAUNCH_K16_TRIM(3) }
        if (a.p1_trim >= 2) NMF_LAUNCH_K16_TRIM(2)
#undef NMF_LAUNCH_K16_TRIM
    }
    if (!wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, false, false, 0, false, OCC>);
    else if (!wstep && partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, false, true, 0, false, OCC>);
    else if (wstep && !partial) NMF_LAUNCH_K16(fused_step_kernel_k16<KT, true, false, 0, false, OCC>);
    else NMF_LAUNCH_K16(fused_step_kernel_k16<KT, true, true, 0, false, OCC>);
#undef NMF_LAUNCH_K16
    return hipGetLastError();
}

template <int KT>
hipError_t launch_check_k16(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream,
                            int batch, size_t strideW, size_t strideH, int nsplit) {
    constexpr int OCC = k16_occ<KT>();
    FusedArgs a;
    a.W = W; a.H = H; a.X = X; a.U_out = nullptr; a.partials = nullptr; a.norm = nullptr;
    a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.Kc = 16 * KT; a.nsplit = nsplit > 0 ? nsplit : 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
    a.strideW = strideW; a.strideH = strideH;
    const size_t lds = k16_lds_bytes<KT>();
    hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_k16<KT, false, false, 0, true, OCC>, lds);
    if (e != hipSuccess) return e;
    note_kernel((const void *)fused_step_kernel_k16<KT, false, false, 0, true, OCC>, stream);
    hipLaunchKernelGGL((fused_step_kernel_k16<KT, false, false, 0, true, OCC>), dim3((unsigned)(((Np + 63) / 64) * a.nsplit), (unsigned)batch), dim3(256), lds, stream, a, part);
    return hipGetLastError();
}

template <int KT>
hipError_t launch_gemm_k16(const float *A, const float *B, float *C, int Mp, int Np, int Kp, hipStream_t stream) {
    constexpr int OCC = k16_occ<KT>();
    FusedArgs a;
    a.W = A; a.H = B; a.X = nullptr; a.U_out = C; a.partials = nullptr; a.norm = nullptr;
    a.Mp = Mp; a.Np = Np; a.Kp = Kp; a.Kc = 16 * KT; a.nsplit = 1; a.partial = 0; a.fast_divide = 0; a.x_in_range = 0;
    const size_t lds = k16_lds_bytes<KT>();
    hipError_t e = ensure_dynamic_lds((const void *)fused_step_kernel_k16<KT, false, false, 0, false, OCC, true>, lds);
    if (e != hipSuccess) return e;
    note_kernel((const void *)fused_step_kernel_k16<KT, false, false, 0, false, OCC, true>, stream);
    hipLaunchKernelGGL((fused_step_kernel_k16<KT, false, false, 0, false, OCC, true>), dim3((Np + 63) / 64), dim3(256), lds, stream, a, (double *)nullptr);
    return hipGetLastError();
}

// Every KT with a kernel: all multiples of 16 from K = 16 to 576 (kMaxK16: round 5 added KT = 33 .. 36 -- 162 KB of the 160 KiB of LDS at
// K = 576, <= 450 registers -- so that K just above 512 stays on this kernel instead of the wave-pair kernel's 77-80 %), in the four groups
// nmf_fused16_inst.hip is compiled in (balanced by code size).  X(KT) is applied to each.
#define NMF_K16_GROUP0(X) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(32) X(17) X(36)
#define NMF_K16_GROUP1(X) X(1) X(11) X(12) X(13) X(14) X(30) X(19) X(31) X(35)
#define NMF_K16_GROUP2(X) X(15) X(16) X(18) X(28) X(21) X(29) X(34)
#define NMF_K16_GROUP3(X) X(20) X(22) X(24) X(26) X(23) X(25) X(27) X(33)
#define NMF_K16_ALL(X) NMF_K16_GROUP0(X) NMF_K16_GROUP1(X) NMF_K16_GROUP2(X) NMF_K16_GROUP3(X)

}  // namespace nmf
