// nmf_kernels.h -- internal launch API of the gfx950 kernels (not part of the C ABI).
// All launchers enqueue on `stream`, never synchronise, never allocate, and return the
// hipError_t of the launch (hipGetLastError), so they are safe inside hipGraph capture.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace nmf {

constexpr float kEps = (float)(2.2204E-16);   // cuda/matrix.cu:10
constexpr int kPad = 32;                      // device buffers are padded to multiples of 32 (cf. PAD_MULT, cuda/matrix.cuh:7)
constexpr int kMaxFusedK = 1024;               // fused path: K <= 1024 (64-column kernel: K <= kMaxK16; wave pairs above)
constexpr int kMaxK16 = 576;                   // the 64-column kernel's largest K: two 32-row LDS images of it + the X patches = 162 KB of 160 KiB

inline int pad32(int v) { return (v + 31) & ~31; }

// Which instantiation did a launcher pick?  With recording on (nmf_debug_record_kernels) every launcher of a fused kernel notes
// the demangled name of the kernel it launches (hipKernelNameRefByPtr of the very function pointer handed to the launch) in a
// thread-local string: tests hit every instantiation BY NAME (tests/test_gpu_instantiations.py).  Off: one relaxed load.
void note_kernel(const void *host_fn, hipStream_t stream);

// ---------------------------------------------------------------- fused half-steps
// One half-step of update_div on zero-padded device buffers (all dims multiples of 32).
//   H-step (wstep=false): streamed factor V = W (Mp x Kp), owned factor U = H (Kp x Np).
//   W-step (wstep=true) : streamed factor V = H,           owned factor U = W.
// Each wave owns 32 columns (H-step) / rows (W-step) of U and streams V in 32-wide chunks
// of the reduction dimension; nsplit > 1 splits that dimension over workgroups and writes
// raw partial products to `partials` (nsplit slabs in U's layout) instead of updating U.
struct FusedArgs {
    const float *W; const float *H; const float *X;   // inputs (W: Mp x Kp, H: Kp x Np, X: Mp x Np)
    float *U_out;             // nsplit == 1: the factor updated in place (H or W)
    float *partials;          // nsplit  > 1: nsplit slabs of Kp*Np (H-step) or Mp*Kp (W-step) floats
    const float *norm;        // Kp clamped normalisers (colsum(W) for the H-step, rowsum(H) for the W-step); nsplit == 1 only
    int Mp, Np, Kp;
    int Kc = 0;               // 16-column kernel: the K its MFMAs cover, a multiple of 16 with Kp == pad32(Kc) (rows / columns Kc .. Kp - 1 of
                              //    the factors are zero padding); 0 = Kp.  The other families compute on Kp.
    int p1_trim = 0;          // 64-column kernel, Kc <= 256: 2 / 3 = the last two / three steps of product 1 cover zero padding
                              //    only (the caller's K <= Kc - 8 / Kc - 12): launch the variant whose chain ends that early (TRIM); 0 = the full chain
    int nsplit;
    int partial;              // 1: write raw partial products (required when nsplit > 1); 0: update U_out in place
    int fast_divide;          // unused since round 5 (the DIV = 1 instantiations are gone: nmf_fused16_impl.h, launch_fused_k16)
    int x_in_range;           // 1: every entry of X is 0 or in [EPS, 2^60] (checked at upload): the 16-column kernel may
                              //    drop the range scaling of IEEE division while the denominators stay <= 2^60 too
    // blockIdx.y = pair of a batched solver (16-column kernel only): W, H of pair b at + b * strideW / strideH floats; its slabs
    // ([nsplit] of them), normalisers (Kp) and vsum_part rows ([nsplit][Kp]) behind those of pair b - 1; the CHECK instantiation's
    // partial triples likewise (nmf_solver_check_all evaluates every pair's check in ONE launch).  batch = gridDim.y.
    size_t strideW = 0, strideH = 0;
    int batch = 1;
    const int *active = nullptr;  // optional [batch] flags: 0 = leave this pair untouched (it has converged)
    float *vsum_part = nullptr;   // optional, W-step with partial slabs on the 16-column kernel (fused_streams_vsum(): Kp <= Mp / 8): nsplit x Kp
                              //    floats receiving, per split, the row sums of the streamed factor H over that split's columns
                              //    -- the W-step normaliser (sum_rows, cuda/nmf.cu:164) read off the LDS image as it streams by
};
hipError_t launch_fused_step(const FusedArgs &a, bool wstep, hipStream_t stream);
// the two kernel families behind launch_fused_step / launch_check (nmf_fused16.hip: 32 <= Kc <= 512, every multiple of 16 up to 256 and
// of 32 above; nmf_fused32.hip: Kp = 32 / 64 / 128 / 256 under NMF_FUSED_VARIANT=3 only)
hipError_t launch_fused16(const FusedArgs &a, bool wstep, hipStream_t stream);
hipError_t launch_fused32(const FusedArgs &a, bool wstep, hipStream_t stream);
// 512 < K <= 1024: two waves share 16 owned columns and split K (nmf_pair16_impl.h); computes on a.Kc = pair_compute_k(K), a multiple
// of 32, with the factors padded to a.Kp = pair_pad_k(K), a multiple of 64, in HBM
hipError_t launch_fused_pair(const FusedArgs &a, bool wstep, hipStream_t stream);
hipError_t launch_check_pair(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc, double *part, hipStream_t stream);
int pair_compute_k(int K);
int pair_pad_k(int K);
// C = A * B through product 1 of the 16-column kernel (the W*H shape: tall A, K <= 512), see nmf_fused16.hip
bool       gemm_nn16_eligible(int m, int n, int k, long lda, long ldb, long ldc);
hipError_t launch_gemm_nn16(const float *A, const float *B, float *C, int Mp, int Np, int Kp, hipStream_t stream);
hipError_t launch_check16(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc, double *part, hipStream_t stream,
                          int batch = 1, size_t strideW = 0, size_t strideH = 0, int nsplit = 1);
int        fused16_compute_k(int K);   // the multiple of 16 the 16-column kernel computes on for K <= kMaxK16, else 0
hipError_t launch_check32(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, double *part, hipStream_t stream,
                          int batch = 1, size_t strideW = 0, size_t strideH = 0);
#ifdef NMF_DIAGNOSTICS   // diagnostic build only (make DIAG=1): not in the shipped library
hipError_t launch_mfma_valu_probe(int nv, int chain, float *out, int iters, hipStream_t stream);   // micro-probe
hipError_t launch_mfma_partner_probe(int mode, float *out, int iters, hipStream_t stream);   // micro-probe 2
hipError_t launch_fused_stamp(const FusedArgs &a, hipStream_t stream);   // diagnostic stamps
hipError_t launch_divide_compare(unsigned long long *counts, unsigned seed, hipStream_t stream);   // diagnostic
hipError_t launch_divide_exhaustive(unsigned long long *counts, int slice, hipStream_t stream);     // diagnostic (slice -1: rcp invariance)
hipError_t launch_fused_probe(const FusedArgs &a, int abl, hipStream_t stream);   // timing probes (v1 kernel + ablation mask)
#endif
// ---------------------------------------------------------------- split half-steps (nmf_split16.hip)
// The half-step for problems whose owned dimension is too short to fill the chip with one workgroup per 64 columns: the
// four waves of a workgroup own the SAME 16 columns and split the reduction dimension (summed through LDS in fixed
// order), the normaliser is summed from the streamed factor in the same pass, and blockIdx.y indexes `batch` independent
// (W, H) pairs against one X.  nsplit == 1: U_out is updated in place by the launch itself.  nsplit > 1: raw slabs
// [batch][nsplit][U] in `partials` plus the per-split sums of the streamed factor [batch][nsplit][Kp] in `vpart`,
// finished by launch_split_apply.
struct SplitArgs {
    const float *W; const float *H; const float *X;   // W: [batch] Mp x Kp, H: [batch] Kp x Np, X: Mp x Np (shared)
    float *U_out;             // nsplit == 1: base of the owned factor (H for the H-step, W for the W-step)
    float *partials;          // nsplit  > 1
    float *vpart;             // nsplit  > 1
    int Mp, Np, Kp;
    int Kc = 0;               // the K the MFMAs cover, a multiple of 16 with Kp == pad32(Kc) (split_compute_k); 0 = Kp
    int Mv, Nv;               // M, N rounded up to 32: column groups beyond them hold only zero padding and get no workgroup
    int nsplit;
    int nw_h, nw_w;           // waves per workgroup (4 or 8) of the H- and the W-step: reduction length % (32 nw) == 0; 8 needs Kp == 64
    int force_partial;        // 1: raw slab + sums even with nsplit == 1 (sharded runs: the all-reduce operand)
    int single_image;         // 64 < Kc <= 128 only: one LDS image instead of two (two workgroups per CU; same arithmetic).  Kc > 128 always runs so
    int batch;
    size_t strideW, strideH;  // floats between consecutive pairs
    const int *active;        // optional [batch] flags: 0 = leave this pair untouched (it has converged)
    int fast_divide, x_in_range;
};
bool       split_step_supports(int Kp);
int        split_compute_k(int K);   // the multiple of 16 (>= 32) the split kernel computes on for K <= 256, else 0
size_t     split_step_lds_bytes(int Kp, int nw, bool double_buffered);
hipError_t launch_split_step(const SplitArgs &a, bool wstep, hipStream_t stream);
// q_valid: rows of W (W-step) / columns of H (H-step) that got a workgroup (SplitArgs::Mv / Nv); the rest is zero padding
hipError_t launch_split_apply(float *U, const float *partials, const float *vpart, int nsplit, int Mp, int Np, int Kp, int q_valid, bool wstep,
                              int batch, size_t ustride, const int *active, hipStream_t stream);

// U[k,q] *= (sum_s partials[s][k,q]) / norm[k]   (col_div/row_div + vec_mul, cuda/matrix.cu:174-250)
// norm == nullptr (W-step only): the normaliser is max(sum_s vsum_part[s][k], EPS) instead
// batch > 1: pair b's factor at U + b * ustride, its slabs, normalisers and vsum_part rows behind those of pair b - 1 (FusedArgs)
hipError_t launch_apply_partials(float *U, const float *partials, int nsplit, const float *norm,
                                 int Mp, int Np, int Kp, bool wstep, hipStream_t stream, const float *vsum_part = nullptr,
                                 int batch = 1, size_t ustride = 0, const int *active = nullptr);
// W-step apply that also leaves norm_out[k] = max(colsum(W_new)[k], EPS), the next H-step's normaliser (sum_cols + set_epsilon,
// cuda/nmf.cu:134-135): one 1024-thread workgroup per column of W.  Normaliser from norm or vsum_part as in launch_apply_partials.
hipError_t launch_apply_w_colsum(float *W, const float *partials, int nsplit, const float *norm, const float *vsum_part,
                                 int Mp, int Kp, float *norm_out, hipStream_t stream, int batch = 1, size_t wstride = 0, const int *active = nullptr);
// psum = sum_s partials[s]   (sharded W-step: operand of the all-reduce)
// vsum_part != nullptr: also psum[count + k] = sum_s vsum_part[s][k], k < Kp (the unclamped row sums of H behind the slab sum)
// q_valid > 0 (with Mp, the slabs' leading dimension): rows >= q_valid of every column were written by no workgroup (zero
// padding) and may hold stale values of another use of the slab buffer: psum gets 0 there without reading them
hipError_t launch_sum_partials(float *psum, const float *partials, int nsplit, size_t count, hipStream_t stream,
                               const float *vsum_part = nullptr, int Kp = 0, int Mp = 0, int q_valid = 0);
// W[m,k] *= psum[m,k] / max(hsum[k], EPS)
hipError_t launch_apply_w(float *W, const float *psum, const float *hsum, int Mp, int Kp, hipStream_t stream);

// The convergence check (reduce1d_div / reduce1d_diff, cuda/matrix.cu:505-640) in one log per element:
//     KL = sum x (log x - log y) - x + y  =  [sum x log x - x]  -  [sum x log y]  +  [sum y],      y = max(W*H, EPS)
//   * sum x log x - x and sum |x| depend on X alone: launch_x_consts, once per upload, fp64;
//   * sum x log y and sum |x - y| need W*H: launch_check (the half-step kernels in CHECK mode: product 1 only, W*H never
//     materialised), per-workgroup partial triples {sum x log2 y, sum |x - y|, 0} in `part`;
//   * sum y = sum_n sum_k colsum(W)_k H[k,n] needs no pass over M x N: launch_check_compose sums the factors in fp64 (the
//     three terms cancel to 1e-2..1e-3 of their size, so fp32 normalisers would cost digits), adds the partials in fixed
//     order and leaves {KL, sum|X-WH|, sum|X|} in out3.
int        check_num_groups(int Np, int Kp);
bool       fused_streams_vsum(int Mp, int Kp);   // can the W-step kernel produce FusedArgs::vsum_part for this shape?
bool       fused_takes_batch(int Kp);   // can a batched solver run this K on the 16-column kernel (blockIdx.y = pair)?
int        fused_cols_per_group(int Kp);   // owned columns per workgroup of the fused kernels (128, or 64 above K = 256)
int        fused_pad_k(int K);             // K as the fused path pads it in HBM (a multiple of 32; of 64 above kMaxK16), 0 if the fused path cannot take it
int        fused_compute_k(int K);         // K as the chosen fused kernel computes on it (FusedArgs::Kc): <= fused_pad_k(K), a multiple of 16
// batch > 1: `batch` (W, H) pairs, strideW / strideH floats apart, in one launch (grid.y); pair b's check_num_groups triples
// at part + 3 * check_num_groups * b.  Kp <= 512 (the batched solvers' range); the wave-pair kernel takes one pair at a time.
// nsplit > 1 (16-column kernel only): the check's reduction over M cut like an H-step's, nsplit workgroups per 64 columns and as many
// triples per pair (check_num_groups * nsplit) -- a 350-column problem is six workgroups otherwise
hipError_t launch_check(const float *W, const float *H, const float *X, int Mp, int Np, int Kp, int Kc,
                        double *part, hipStream_t stream, int batch = 1, size_t strideW = 0, size_t strideH = 0, int nsplit = 1);
hipError_t launch_check_final(const double *part, int ngroups, double *out3, hipStream_t stream);
constexpr int kXConstGroups = 1024;     // partial triples of launch_x_consts
constexpr int kSum64Blocks = 2048;      // workgroup partials of the weighted fp64 sum over H
// xc3 <- {sum x log x - x, sum |x|, 0} over the n floats of X (zeros skipped); part: 3 * kXConstGroups doubles of scratch
hipError_t launch_x_consts(const float *X, size_t n, double *part, double *xc3, hipStream_t stream);
// scratch64: Kp + kSum64Blocks doubles
hipError_t launch_check_compose(const double *part, int ngroups, const float *W, const float *H, int Mp, int Np, int Kp,
                                const double *xc3, double *scratch64, double *out3, hipStream_t stream);

// ---------------------------------------------------------------- normalisers
// out[k] = max(sum_i A[i + k*ld], EPS), one workgroup per column (sum_cols + set_epsilon, cuda/nmf.cu:134-135)
// batch > 1: matrix b at A + b * astride, its sums at out + b * cols
hipError_t launch_col_sums(const float *A, int rows, int cols, long ld, float *out, bool clamp, hipStream_t stream, int batch = 1, size_t astride = 0);
// row sums in two deterministic levels: part[b][k] then out[k] = max(sum_b part[b][k], EPS) (cuda/nmf.cu:164-165)
int        row_sum_blocks(int cols);
// batch > 1: matrix b at A + b * astride, its block partials and sums behind those of matrix b - 1
hipError_t launch_row_sums(const float *A, int rows, int cols, long ld, float *part, float *out, bool clamp,
                           hipStream_t stream, int batch = 1, size_t astride = 0);

// ---------------------------------------------------------------- unfused operators
enum GemmKind { GEMM_NN = 0, GEMM_TN = 1, GEMM_NT = 2 };
// C(m x n, ldc) = op(A) * op(B) with reduction length k; fp32 MFMA, arbitrary sizes.
// `workspace` (optional): enables split-K for small outputs with a long reduction (slabs summed in fixed order).
hipError_t launch_gemm(GemmKind kind, int m, int n, int k, const float *A, long lda, const float *B, long ldb,
                       float *C, long ldc, hipStream_t stream, float *workspace = nullptr, size_t workspace_floats = 0);
hipError_t launch_set_epsilon(float *a, size_t n, hipStream_t stream);
hipError_t launch_vec_div(const float *a, const float *b, float *c, size_t n, hipStream_t stream);
hipError_t launch_vec_mul(const float *a, const float *b, float *c, size_t n, hipStream_t stream);
// c[i + j*ld] = a[i + j*ld] / b[i]   (col_div, cuda/matrix.cu:244-250)
hipError_t launch_col_div(const float *a, const float *b, float *c, int rows, int cols, long ld, hipStream_t stream);
// c[i + j*ld] = a[i + j*ld] / b[j]   (row_div, cuda/matrix.cu:220-224)
hipError_t launch_row_div(const float *a, const float *b, float *c, int rows, int cols, long ld, hipStream_t stream);
// generic KL / diff reductions over two flat arrays (reduce1d_div / reduce1d_diff): 3 doubles per group
int        reduce_num_groups(size_t n);
hipError_t launch_kl_reduce(const float *x, const float *y, size_t n, double *part, hipStream_t stream);

// zero `bytes` (a multiple of 16, 16-byte aligned) with 16-byte stores: a solver's arena at creation.  hipMemsetAsync took
// 18-21 ms for the 12-49 MiB of the gold shape's buffers (a byte-wise path: ~2.5 GB/s) where this takes ~0.03 ms
hipError_t launch_zero(void *p, size_t bytes, hipStream_t stream);

// ---------------------------------------------------------------- padding helpers
// dst (rows_p x cols_p, ld = rows_p) <- src (rows x cols, ld = rows), zero padding, optional EPS clamp of the
// valid region (read_matrix's set_epsilon, cuda/nmf.cu:211)
// range_flag (may be null): set to 1 if any copied value is NaN or > 2^60
hipError_t launch_pad_copy(float *dst, int rows_p, int cols_p, const float *src, int rows, int cols, bool clamp, unsigned *range_flag,
                           hipStream_t stream);
hipError_t launch_unpad_copy(float *dst, int rows, int cols, const float *src, int rows_p, hipStream_t stream);

}  // namespace nmf
