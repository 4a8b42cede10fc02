// nmf_split16_impl.h -- the half-step of update_div (cuda/nmf.cu:118-176) for problems that do not fill 256 CUs with one
// workgroup per 64 owned columns: the reference's own workload (4096 x 350 x 128, matrix_export.py:4-7), BASELINE config 2
// (1024 x 4096 x 64), the paper's 512 x 3445 x 30, and B restarts of any of them in one launch (paper section 3.2).
// Included by nmf_split16.hip (dispatch, apply kernel) and nmf_split16_inst.hip (the instantiations, compiled in groups).
//
// Same arithmetic as fused_step_kernel_k16 (nmf_fused16_impl.h) -- v_mfma_f32_16x16x4_f32, 16 owned columns per wave, the
// quotient fed from the accumulator of product 1 straight into product 2 -- with a different division of labour:
//   * the FOUR WAVES OF A WORKGROUP OWN THE SAME 16 COLUMNS and split the reduction dimension between them: the
//     workgroup stages a "superchunk" of 128 rows of the streamed factor as four 32-row sub-images and wave w works on
//     sub-image w.  The four K x 16 accumulators are summed through LDS in a fixed order at the end, and the update
//     U *= Acc / norm is applied by the same launch: a problem with Q/16 >= ~256 column groups needs no partial slabs and
//     no apply kernel at all (cfg2's H-step, the gold shape's W-step).
//   * the normaliser (colsum(W) for the H-step, rowsum(H) for the W-step: sum_cols / sum_rows + set_epsilon,
//     cuda/nmf.cu:134-135,164-165) is the sum of the streamed factor, which every workgroup stages anyway: each thread
//     adds the 16-byte pieces it stages to a private accumulator (4 VALU adds per piece) and one LDS reduction at the end
//     yields all K sums.  No col_sums / row_sums launch, no hand-over between kernels: the normaliser is a function of the
//     factor the kernel reads, so a resumed run equals an uninterrupted one by construction.
//   * where Q/16 is small the reduction dimension is also split over workgroups (nsplit > 1): raw slabs + per-split sums
//     of the streamed factor, finished by split_apply_kernel.
//   * blockIdx.y is a restart index: B (W, H) pairs against one resident X, whose tiles the restarts share through L2.
#pragma once
#include "nmf_device.h"

#include <type_traits>

namespace nmf {

constexpr int kXs16Floats = 32 * 20;   // per-wave X patch: H-step 16 x 36, W-step 32 x 20 floats

// NW = waves per workgroup = 32-row sub-images per superchunk (4, or 8 at K = 64: one workgroup then puts two waves on
// every SIMD of its CU, and each fills the other's quotient / wait / barrier time with MFMAs -- what a second workgroup
// per CU does for the 64-column kernel, for shapes with no more than one workgroup per CU to hand out).
// KT = K / 16 (2 .. 16): like the 64-column kernel the split kernel exists for every multiple of 16 -- the reference pads K
// to 32 and nothing coarser (cuda/matrix.cuh:7), and the many NMF problems with a few dozen components (the paper's R = 30
// among them) should not pay for zeros.  The factors in HBM and the LDS images are padded to KS = 32 ceil(KT / 2) columns
// (= SplitArgs::Kp); the MFMAs cover K = 16 KT (= SplitArgs::Kc).  k index of product-1 step s: k16_kconst (nmf_device.h).
// DB = false: ONE superchunk image in LDS instead of two.  The next image then cannot be written behind product 2; it goes
// to LDS between two barriers after it (from the registers the loads of this iteration filled), which exposes ~4 NPC ds_writes
// per superchunk -- and halves the LDS footprint: 64 < K <= 128 fits two workgroups per CU (<= 78 KiB each), which is what a
// batch of restarts needs to overlap one workgroup's quotient / barrier / first-touch time with another's MFMAs, and K > 128 fits
// at all (<= 135 KiB).  Same arithmetic in the same order as DB = true: the choice may depend on the batch size.
template <int KT, int NW, bool WSTEP, bool PARTIAL, int DIV, int OCC, bool DB = true>
__global__ __launch_bounds__(64 * NW, OCC) void split_step_kernel_k16(SplitArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int K = 16 * KT;       // what the MFMAs cover (SplitArgs::Kc)
    constexpr int NST = (KT + 1) / 2;   // staged 16-B pieces per thread per 32-row chunk: 32 columns of the streamed factor each
    constexpr int KS = 32 * NST;     // staged columns = SplitArgs::Kp (zero padding beyond K)
    constexpr int VBUF = KS * kLdv;  // one 32-wide sub-image
    constexpr int IMG = NW * VBUF;   // one superchunk
    constexpr int SH = NW / 4;       // sub-images staged per pass of the workgroup's 8 NW piece slots (32 slots per sub-image)
    constexpr int N1 = 4 * KT;       // product-1 steps per 16-row tile
    constexpr int NT = KT;           // 16 x 16 accumulator tiles
    constexpr int NPC = 4 * NST;     // staged pieces per thread per superchunk
    constexpr int NF = 16 * (KT / 4);   // product-1 steps in whole 64-blocks of k
    constexpr int RR = (K % 64) / 4;    // product-1 steps in the remainder block (0, 4, 8, 12): step s' covers k = 64 (K / 64) + 4 s' + kq
    constexpr int D = kRing < 2 * N1 ? kRing : 2 * N1;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 15, kq = lane >> 4;
    // One wave per SIMD has nobody to hide anything behind: the 64-column kernel's first, run-time form of product-1 trimming (a uniform
    // branch in front of the chain's tail + compiler-placed MFMAs) measured -3 % here with three steps skipped and -9 .. -12 % with none
    // (profiles/r04_small_levers.log); this kernel computes on the next multiple of 16 and keeps the run map in the remainder block
    // (conflict-free at K % 64 = 32) and a straight-line inline-asm chain.
    constexpr bool IL = false;
    const int rl = k16_rem_lane<RR, IL>(kq);   // lane part of the k index in the remainder block
    const int b = blockIdx.y;
    if (a.active != nullptr && a.active[b] == 0) return;
    const int P = WSTEP ? a.Np : a.Mp;   // a multiple of 32 NW: every superchunk is whole
    const int nsplit = a.nsplit;
    const bool x_in_range = a.x_in_range != 0;
    int split, qblk;
    if (!WSTEP) { split = blockIdx.x % nsplit; qblk = blockIdx.x / nsplit; }
    else {
        // Two neighbouring row groups share every 128-byte line of X (16 rows = 64 bytes): give both to the same XCD
        // (workgroups id and id + 8 share one under round-robin dispatch), so that the second one hits in that XCD's L2.
        const unsigned id = blockIdx.x, whole = gridDim.x & ~15u;
        const unsigned L = id < whole ? (id & ~15u) + ((id & 7u) << 1) + ((id >> 3) & 1u) : id;
        const unsigned m = L >> 1;
        split = (int)(m % (unsigned)nsplit);
        qblk = 2 * (int)(m / (unsigned)nsplit) + (int)(L & 1u);
    }
    const int q0 = qblk * 16;
    const float *__restrict__ Wb = a.W + (size_t)b * a.strideW;
    const float *__restrict__ Hb = a.H + (size_t)b * a.strideH;
    const float *__restrict__ V = WSTEP ? Hb : Wb;
    const float *__restrict__ U = WSTEP ? Wb : Hb;
    const long ldv = WSTEP ? a.Kp : a.Mp, ldu = WSTEP ? a.Mp : a.Kp, ldx = a.Mp;
    const int nsc = P / (32 * NW);
    const int scps = (nsc + nsplit - 1) / nsplit;
    const int sc_begin = split * scps;
    const int sc_end = (sc_begin + scps < nsc) ? (sc_begin + scps) : nsc;

    // B operands of product 1: ub[s] = U(k(s, kq), q0 + j)
    float ub[N1];
    if (!WSTEP) {
        const float *__restrict__ col = U + (size_t)(16 * kq) + (size_t)(q0 + j) * ldu;
#pragma unroll
        for (int sb = 0; sb < KT / 4; ++sb)
#pragma unroll
            for (int e4 = 0; e4 < 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(col + 64 * sb + 4 * e4);
                ub[16 * sb + 4 * e4] = v[0]; ub[16 * sb + 4 * e4 + 1] = v[1]; ub[16 * sb + 4 * e4 + 2] = v[2]; ub[16 * sb + 4 * e4 + 3] = v[3];
            }
        if (RR > 0) {   // the remainder block: a run of RR (a multiple of 4: 16-B aligned) at 64 (K / 64) + RR pi(kq)
            const float *__restrict__ colr = U + (size_t)(64 * (KT / 4) + rl) + (size_t)(q0 + j) * ldu;
#pragma unroll
            for (int e4 = 0; e4 < RR / 4; ++e4) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(colr + 4 * e4);
                ub[NF + 4 * e4] = v[0]; ub[NF + 4 * e4 + 1] = v[1]; ub[NF + 4 * e4 + 2] = v[2]; ub[NF + 4 * e4 + 3] = v[3];
            }
        }
    } else {
#pragma unroll
        for (int s = 0; s < N1; ++s) ub[s] = U[(size_t)(q0 + j) + (size_t)(k16_kconst<NF, IL>(s) + (k16_in_rem<NF>(s) ? rl : 16 * kq)) * ldu];
    }

    // The values this wave will update in the epilogue (tiles t = wave, wave + NW, ...: U(16 t + 4 kq + r, q0 + j)), fetched
    // now so that the read-modify-write at the end does not wait for a global round trip on an otherwise idle CU
    constexpr int NTW = (NT + NW - 1) / NW;
    f32x4 uold[NTW];
    if (!PARTIAL) {
        const float *__restrict__ Ub = U;
#pragma unroll
        for (int tt = 0; tt < NTW; ++tt) {
            const int t = wave + NW * tt;
            const int k = 16 * (t < NT ? t : 0) + 4 * kq;
            if (!WSTEP) uold[tt] = *reinterpret_cast<const f32x4 *>(Ub + (size_t)k + (size_t)(q0 + j) * ldu);
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) uold[tt][r] = Ub[(size_t)(q0 + j) + (size_t)(k + r) * ldu];
            }
        }
    }

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // sums of the streamed factor over the pieces this thread stages.  H-step: piece = 4 consecutive rows p of column
    // k = g + 32 qq of W; W-step: piece = rows k = 4 (8 qq + (tid & 7)) .. + 3 of column p = g of H, where g = (tid >> 3) & 31
    // and pass sp of the staging puts this thread on sub-image SH sp + (tid >> 8).
    const int g = (tid >> 3) & 31, sub_hi = tid >> 8;
    // PARTIAL: the sums go to vpart, which only the workgroups of the first column group write
    const bool vs_on = !PARTIAL || qblk == 0;
    f32x4 vs[NST];
#pragma unroll
    for (int q = 0; q < NST; ++q) vs[q] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (sc_begin < sc_end) {
        const unsigned vstep = 4u * (WSTEP ? 32u : 32u * (unsigned)ldv);              // bytes between the pieces of one chunk
        const size_t vchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldv : (size_t)32);    // bytes between chunks
        const unsigned voff0 = 4u * ((unsigned)(4 * (tid & 7)) + (unsigned)g * (unsigned)ldv) + (unsigned)sub_hi * (unsigned)vchunk;
        // X tile (32 p x 16 q): H-step 16 columns of 128 B (8 lanes per column), W-step 32 rows of 64 B (4 lanes per row)
        const unsigned xoff0 = WSTEP ? 4u * ((unsigned)(4 * (lane & 3)) + (unsigned)(lane >> 2) * (unsigned)ldx)
                                     : 4u * ((unsigned)(4 * (lane & 7)) + (unsigned)(lane >> 3) * (unsigned)ldx);
        const unsigned xstep = WSTEP ? 4u * 16u * (unsigned)ldx : 4u * 8u * (unsigned)ldx;
        const char *__restrict__ xbase = reinterpret_cast<const char *>(WSTEP ? a.X + (size_t)q0 : a.X + (size_t)q0 * (size_t)ldx);
        const size_t xchunk = 4 * (WSTEP ? (size_t)32 * (size_t)ldx : (size_t)32);
        // The X tile of superchunk s is loaded while s - 2 is being computed (one wave per SIMD has nobody to hide an L2
        // miss behind), parked in the wave's LDS patch at the end of that iteration, and read back in the accumulator
        // layout during s - 1, after that iteration's quotient and before its own parking.
        float *xt = smem + (DB ? 2 : 1) * IMG + wave * kXs16Floats;   // no __restrict__: written and read back within the wave
        const int xw_off = WSTEP ? (lane >> 2) * 20 + 4 * (lane & 3) : (lane >> 3) * kXtLd + 4 * (lane & 7);
        const int xr_off = WSTEP ? 4 * kq * 20 + j : j * kXtLd + 4 * kq;
        const int p1_off = 16 * kq * kLdv + j;     // + k16_kconst(s) * kLdv + 16 T
        const int p1r_off = rl * kLdv + j;         // the same for the steps of the remainder block
        const int p2_off = j * kLdv + 4 * kq;      // + 16 t * kLdv + 16 T + r

        f32x4 st[NPC];
        f32x4 xg[2];
        float xr[8];
        unsigned vo = voff0, xo = xoff0;
        const char *__restrict__ vcur = reinterpret_cast<const char *>(V);
        const char *__restrict__ xcur = xbase;
        auto set_v = [&](int sc) {
            vo = voff0;
            asm volatile("" : "+v"(vo));
            vcur = reinterpret_cast<const char *>(V) + (size_t)(NW * sc) * vchunk;
        };
        auto set_x = [&](int sc) {
            xo = xoff0;
            asm volatile("" : "+v"(xo));
            xcur = xbase + (size_t)(NW * sc + wave) * xchunk;
        };
        auto stage_load_one = [&](int w) {   // piece w = (pass sp, qq)
            const int sp = w / NST, qq = w % NST;
            global_bytes base = (global_bytes)(vcur + (size_t)(SH * sp) * vchunk + (size_t)qq * (size_t)vstep);
            asm volatile("" : "+s"(base));
            st[w] = *(const __attribute__((address_space(1))) f32x4 *)(base + vo);
        };
        auto x_load_one = [&](int i) {
            global_bytes base = (global_bytes)(xcur + (size_t)i * (size_t)xstep);
            asm volatile("" : "+s"(base));
            xg[i] = *(const __attribute__((address_space(1))) f32x4 *)(base + xo);
        };
        const int img_off = sub_hi * VBUF + (WSTEP ? 4 * (tid & 7) * kLdv + g : g * kLdv + 4 * (tid & 7));
        auto vs_add = [&](int w) {   // piece w joins this thread's share of the streamed factor's sums
            const int qq = w % NST;
            if (!WSTEP) vs[qq][0] += (st[w][0] + st[w][1]) + (st[w][2] + st[w][3]);
            else        vs[qq] += st[w];
        };
        auto stage_store_one = [&](float *__restrict__ img, int w4, auto vs_tag) {   // one ds_write_b32 of piece w4 / 4
            const int w = w4 / 4, cc = w4 % 4, sp = w / NST, qq = w % NST;
            float *__restrict__ vl = img + img_off + SH * sp * VBUF;
            if (!WSTEP) vl[32 * qq * kLdv + cc] = st[w][cc];
            else        vl[(32 * qq + cc) * kLdv] = st[w][cc];
            if (decltype(vs_tag)::value && cc == 3) vs_add(w);   // the piece is complete
        };
        auto x_park = [&]() {    // xg -> patch
            float *w = xt + xw_off;
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(w + (WSTEP ? 16 * i * 20 : 8 * i * kXtLd)) = xg[i];
        };
        auto x_fetch = [&]() {   // patch -> xr, the layout of product 1's result
            const float *r = xt + xr_off;
            if (!WSTEP) {
#pragma unroll
                for (int T = 0; T < 2; ++T) {
                    const f32x4 v = *reinterpret_cast<const f32x4 *>(r + 16 * T);
                    xr[4 * T] = v[0]; xr[4 * T + 1] = v[1]; xr[4 * T + 2] = v[2]; xr[4 * T + 3] = v[3];
                }
            } else {
#pragma unroll
                for (int T = 0; T < 2; ++T)
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) xr[4 * T + r4] = r[(16 * T + r4) * 20];
            }
        };

        const int sc_last = sc_end - 1;
        // Prologue: every global load of the first superchunk and both leading X tiles is issued before the first wait, so the
        // kernel pays ONE first-touch round trip (the factor was written by the previous launch, on other XCDs), not two
        f32x4 xg0[2];
        set_x(sc_begin);
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // the first tile into registers of its own: no copy, hence no wait, before the next loads
            global_bytes base = (global_bytes)(xcur + (size_t)i * (size_t)xstep);
            asm volatile("" : "+s"(base));
            xg0[i] = *(const __attribute__((address_space(1))) f32x4 *)(base + xo);
        }
        set_x(sc_begin + 1 < sc_end ? sc_begin + 1 : sc_last);
#pragma unroll
        for (int i = 0; i < 2; ++i) x_load_one(i);
        set_v(sc_begin);
#pragma unroll
        for (int w = 0; w < NPC; ++w) stage_load_one(w);
        {
            float *w = xt + xw_off;
#pragma unroll
            for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4 *>(w + (WSTEP ? 16 * i * 20 : 8 * i * kXtLd)) = xg0[i];
        }
        x_fetch();
        x_park();
#pragma unroll
        for (int w4 = 0; w4 < 4 * NPC; ++w4) stage_store_one(smem, w4, std::true_type{});
        __syncthreads();

        // one superchunk; LAST: nothing left to stage (the peeled final iteration); VS: this workgroup's sums are needed
        auto body = [&](int sc, auto last_tag, auto vs_tag) {
            constexpr bool LAST = decltype(last_tag)::value;
            const int rel = sc - sc_begin, par = rel & 1;
            const float *__restrict__ vb = smem + (DB ? par * IMG : 0) + wave * VBUF;
            float *__restrict__ vn = smem + (DB ? (par ^ 1) * IMG : 0);
            // The sums of the image staged behind the previous iteration's product 2 are taken here, ahead of the MFMA chain: a
            // VALU add costs ~4 cycles while nothing else is in flight and ~10 between two f32 MFMAs (stamps: 340 cycles per
            // superchunk in product 2); the registers still hold the pieces until this iteration's loads overwrite them.  Same
            // pieces in the same order as before: the sums are bit-identical.
            if (DB && decltype(vs_tag)::value && rel > 0) {
#pragma unroll
                for (int w = 0; w < NPC; ++w) vs_add(w);
            }
            if (!LAST) { set_v(sc + 1); set_x(sc + 2 < sc_end ? sc + 2 : sc_last); }
            // ---- product 1: two interleaved chains, step index e = 2 s + T
            const lds_float *b1 = (const lds_float *)vb + p1_off;
            const lds_float *b1r = (const lds_float *)vb + p1r_off;
            auto a1_ld = [&](int e) { return lds_ld((k16_in_rem<NF>(e >> 1) ? b1r : b1) + k16_kconst<NF, IL>(e >> 1) * kLdv + 16 * (e & 1)); };
            constexpr int E1 = 2 * N1;
            float ar[D];
#pragma unroll
            for (int e = 0; e < D; ++e) ar[e] = a1_ld(e);
            if (OCC > 1) __builtin_amdgcn_s_setprio(1);
            f32x4 s0, s1;
            constexpr int NLOAD = NPC + 2;
            constexpr int G = E1 / (NLOAD + 1);
#pragma unroll
            for (int e = 0; e < E1; ++e) {
                const int s = e >> 1;
                if (KT > 8) {
                    // K > 128 (one workgroup per CU, one LDS image) keeps part of its operands in AGPRs; the copies the compiler makes for an
                    // inline-asm MFMA sit right in front of it, and the hazard recogniser cannot see into the asm (measured: ~1 % wrong sums in
                    // the W-step).  The builtin is an instruction the compiler knows: it places the wait states itself.
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
                if (!LAST && e >= G && e % G == 0 && e / G - 1 < NLOAD) {
                    const int l = e / G - 1;
                    if (l < NPC) stage_load_one(l); else x_load_one(l - NPC);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            // ---- first operands of product 2, then the quotient in one VALU block
            const lds_float *b2 = (const lds_float *)vb + p2_off;
            constexpr int E2 = 8 * NT;   // order: (T, r) outer, tile t inner
            float a2[D];
#pragma unroll
            for (int e = 0; e < D; ++e) a2[e] = lds_ld(b2 + 16 * (e % NT) * kLdv + 16 * ((e / NT) >> 2) + ((e / NT) & 3));
            float z[8];
            // An MFMA's result needs 12 wait states (8 passes + 4) before anything but an accumulating MFMA touches it, and the compiler
            // pads nothing behind an asm MFMA: the D = 8 LDS reads above count, six nops make up the rest (tools/asm_audit.py checks).
            static_assert(D == 8, "the wait states behind product 1 count on eight ds_reads");
            asm volatile("s_nop 5" : "+v"(s0), "+v"(s1));
            __builtin_amdgcn_sched_barrier(0);
            quotient8<DIV>(xr, s0, s1, z, x_in_range);
            __builtin_amdgcn_sched_barrier(0);
            if (!LAST) x_fetch();   // the tile of superchunk sc + 1
            if (OCC > 1) __builtin_amdgcn_s_setprio(1);
            // ---- product 2: NT independent accumulators; the next superchunk's image goes to LDS one ds_write per MFMA
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
                    if (DB && !LAST && e < 4 * NPC) {   // one per MFMA: all 4 NPC of them where KT is even (4 NPC == E2)
                        stage_store_one(vn, e, std::false_type{});
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            if (DB && !LAST) {   // odd KT: the image is 32 ceil(KT / 2) columns wide, eight ds_writes more than there are MFMAs in product 2
#pragma unroll
                for (int w4 = E2; w4 < 4 * NPC; ++w4) stage_store_one(vn, w4, std::false_type{});
            }
            if (OCC > 1) __builtin_amdgcn_s_setprio(0);
            if (!LAST) x_park();    // the tile of superchunk sc + 2
            __syncthreads();
            if (!DB && !LAST) {     // everybody has finished with the one image: replace it
#pragma unroll
                for (int w4 = 0; w4 < 4 * NPC; ++w4) stage_store_one(vn, w4, vs_tag);
                __syncthreads();
            }
        };
        if (vs_on) { for (int sc = sc_begin; sc < sc_last; ++sc) body(sc, std::false_type{}, std::true_type{}); body(sc_last, std::true_type{}, std::true_type{}); }
        else       { for (int sc = sc_begin; sc < sc_last; ++sc) body(sc, std::false_type{}, std::false_type{}); body(sc_last, std::true_type{}, std::false_type{}); }
    }

    // ---- the four waves' accumulators -> one (fixed order), the streamed factor's sums -> K normalisers
    // red[w][t][lane] (f32x4) at smem, vsl[slot][k] behind it; the superchunk images are dead (barrier above)
    f32x4 *red = reinterpret_cast<f32x4 *>(smem);
    float *vsl = smem + NW * NT * 64 * 4;
    constexpr int NSLOT = WSTEP ? 8 * NW : 8 * SH;
    float *nrm_l = vsl + NSLOT * KS;   // all KS staged columns are summed (the padding columns to zero): vpart / the normalisers have SplitArgs::Kp entries
#pragma unroll
    for (int t = 0; t < NT; ++t) red[(wave * NT + t) * 64 + lane] = acc[t];
    if (!WSTEP) {   // slot = (tid & 7, sub_hi): 8 SH threads share a column k of W
#pragma unroll
        for (int qq = 0; qq < NST; ++qq) vsl[((tid & 7) + 8 * sub_hi) * KS + g + 32 * qq] = vs[qq][0];
    } else {        // slot = tid >> 3: 8 NW threads share four rows k of H
#pragma unroll
        for (int qq = 0; qq < NST; ++qq)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) vsl[(tid >> 3) * KS + 4 * (8 * qq + (tid & 7)) + cc] = vs[qq][cc];
    }
    __syncthreads();
    if (tid < KS) {
        float n = vsl[tid];
#pragma unroll
        for (int sl = 1; sl < NSLOT; ++sl) n += vsl[sl * KS + tid];
        if (PARTIAL) { if (qblk == 0) a.vpart[((size_t)b * nsplit + split) * KS + tid] = n; }
        else nrm_l[tid] = clamp_eps(n);
    }
    if (!PARTIAL) __syncthreads();
    // wave w finishes tiles t = w, w + NW, ...: lane holds Acc(k = 16 t + 4 kq + r, q0 + j)
    const size_t ustride = WSTEP ? a.strideW : a.strideH;
#pragma unroll
    for (int tt = 0; tt < (NT + NW - 1) / NW; ++tt) {
        const int t = wave + NW * tt;
        if (t >= NT) break;
        const f32x4 r0 = red[(0 * NT + t) * 64 + lane], r1 = red[(1 * NT + t) * 64 + lane], r2 = red[(2 * NT + t) * 64 + lane], r3 = red[(3 * NT + t) * 64 + lane];
        f32x4 sum = (r0 + r1) + (r2 + r3);
        if (NW == 8) {
            const f32x4 r4 = red[(4 * NT + t) * 64 + lane], r5 = red[(5 * NT + t) * 64 + lane], r6 = red[(6 * NT + t) * 64 + lane], r7 = red[(7 * NT + t) * 64 + lane];
            sum += (r4 + r5) + (r6 + r7);
        }
        const int k = 16 * t + 4 * kq;
        if (PARTIAL) {
            const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
            float *__restrict__ out = a.partials + ((size_t)b * nsplit + split) * slab;
            if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)k + (size_t)(q0 + j) * ldu) = sum;
            else {
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(k + r) * ldu] = sum[r];
            }
        } else {
            float *__restrict__ Uo = a.U_out + (size_t)b * ustride;
            if (!WSTEP) {
                float *p = Uo + (size_t)k + (size_t)(q0 + j) * ldu;
                f32x4 u = uold[tt];
#pragma unroll
                for (int e = 0; e < 4; ++e) u[e] = u[e] * (sum[e] / nrm_l[k + e]);
                *reinterpret_cast<f32x4 *>(p) = u;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float *p = Uo + (size_t)(q0 + j) + (size_t)(k + r) * ldu;
                    *p = uold[tt][r] * (sum[r] / nrm_l[k + r]);
                }
            }
        }
    }
    if ((KT & 1) && PARTIAL && wave == KT % NW) {
        // rows K .. KS - 1 of the slab are zero padding: written, because the slab buffer is shared between the half-steps and what
        // its consumers multiply into the (zero) padding of U must at least be finite
        const size_t slab = WSTEP ? (size_t)a.Mp * a.Kp : (size_t)a.Kp * a.Np;
        float *__restrict__ out = a.partials + ((size_t)b * nsplit + split) * slab;
        if (!WSTEP) *reinterpret_cast<f32x4 *>(out + (size_t)(K + 4 * kq) + (size_t)(q0 + j) * ldu) = f32x4{0.f, 0.f, 0.f, 0.f};
        else {
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(size_t)(q0 + j) + (size_t)(K + 4 * kq + r) * ldu] = 0.f;
        }
    }
}

// The launcher of one (KT, NW, OCC, DB): defined here, instantiated explicitly in nmf_split16_inst.hip (one group of KT values per
// compilation), declared `extern template` where it is called from (nmf_split16.hip).
template <int KT, int NW, int OCC, bool DB>
hipError_t launch_split_k16(const SplitArgs &a, bool wstep, hipStream_t stream) {
    const int Q = wstep ? a.Mv : a.Nv;   // the column groups beyond hold zero padding only: it stays zero without being touched
    const dim3 grid((unsigned)((Q / 16) * a.nsplit), (unsigned)a.batch), block(64 * NW);
    const size_t lds = ((size_t)(DB ? 2 : 1) * NW * 32 * ((KT + 1) / 2) * kLdv + NW * kXs16Floats) * sizeof(float);
    const bool partial = a.nsplit > 1 || a.force_partial;
    // (no DIV = 1 instantiations since round 5: nmf_fused16_impl.h, launch_fused_k16)
#define NMF_LAUNCH_S16(...)                                                                               \
    do {                                                                                                  \
        hipError_t e = ensure_dynamic_lds((const void *)__VA_ARGS__, lds);                                \
        if (e != hipSuccess) return e;                                                                    \
        note_kernel((const void *)__VA_ARGS__, stream); \
        hipLaunchKernelGGL((__VA_ARGS__), grid, block, lds, stream, a);                                   \
    } while (0)
    if (!wstep && !partial) NMF_LAUNCH_S16(split_step_kernel_k16<KT, NW, false, false, 0, OCC, DB>);
    else if (!wstep && partial) NMF_LAUNCH_S16(split_step_kernel_k16<KT, NW, false, true, 0, OCC, DB>);
    else if (wstep && !partial) NMF_LAUNCH_S16(split_step_kernel_k16<KT, NW, true, false, 0, OCC, DB>);
    else NMF_LAUNCH_S16(split_step_kernel_k16<KT, NW, true, true, 0, OCC, DB>);
#undef NMF_LAUNCH_S16
    return hipGetLastError();
}

// Every configuration with a kernel, X(KT, NW, OCC, DB), in the four groups nmf_split16_inst.hip is compiled in:
//   K <= 64: two LDS images, two workgroups per CU (and eight waves per workgroup at K = 64: NMF_SPLIT_NW=8);
//   64 < K <= 128: two images at one workgroup per CU, or one image at two per CU (a batch of restarts);
//   K > 128: one image, one workgroup per CU.
#define NMF_S16_GROUP0(X) X(2, 4, 2, true) X(3, 4, 2, true) X(4, 4, 2, true) X(4, 8, 2, true) X(5, 4, 1, true) X(5, 4, 2, false) X(16, 4, 1, false)
#define NMF_S16_GROUP1(X) X(6, 4, 1, true) X(6, 4, 2, false) X(7, 4, 1, true) X(7, 4, 2, false) X(15, 4, 1, false)
#define NMF_S16_GROUP2(X) X(8, 4, 1, true) X(8, 4, 2, false) X(9, 4, 1, false) X(10, 4, 1, false) X(14, 4, 1, false)
#define NMF_S16_GROUP3(X) X(11, 4, 1, false) X(12, 4, 1, false) X(13, 4, 1, false)
#define NMF_S16_ALL(X) NMF_S16_GROUP0(X) NMF_S16_GROUP1(X) NMF_S16_GROUP2(X) NMF_S16_GROUP3(X)

}  // namespace nmf
