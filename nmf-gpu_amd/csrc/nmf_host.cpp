// nmf_host.cpp -- host side of libnmf_mi355x.so: the resident solver, the update_div loop with
// its convergence contract, hipGraph capture/replay, timers, and the per-operator C entry points.
//
// Mirrors (does not translate) the reference's host logic: run_async / update_h / update_w
// (cuda/nmf.cu:76-176) and the operator wrappers of cuda/matrix.cu:97-260.
#include "../../include/nmf_mi355x.h"
#include "nmf_comm.h"
#include "nmf_kernels.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

using namespace nmf;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
static void set_err(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
extern "C" const char *nmf_last_error(void) { return g_err; }
void nmf_internal_set_error(const char *msg) { snprintf(g_err, sizeof g_err, "%s", msg ? msg : ""); }

#define HIPCHK(expr)                                                                                      \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            set_err("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_));                 \
            return NMF_ERR_HIP;                                                                           \
        }                                                                                                 \
    } while (0)
#define NMFCHK(expr)                                                                                      \
    do {                                                                                                  \
        int s_ = (expr);                                                                                  \
        if (s_ != NMF_OK) return s_;                                                                      \
    } while (0)

extern "C" const char *nmf_status_string(int st) {
    switch (st) {
        case NMF_OK: return "ok";
        case NMF_ERR_ARG: return "invalid argument";
        case NMF_ERR_SHAPE: return "dimensions do not agree";
        case NMF_ERR_HIP: return "HIP runtime error";
        case NMF_ERR_IO: return "file I/O error";
        case NMF_ERR_NOMEM: return "out of memory";
        case NMF_ERR_COMM: return "RCCL error";
        case NMF_ERR_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown status";
    }
}
extern "C" const char *nmf_version(void) { return "nmf_mi355x 0.5 (gfx950)"; }

extern "C" int nmf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
extern "C" int nmf_device_name(int device, char *buf, int buflen) {
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    snprintf(buf, (size_t)buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return NMF_OK;
}

extern "C" void nmf_default_opts(nmf_opts *o) {
    if (!o) return;
    memset(o, 0, sizeof *o);
    o->converge_thresh = 0.f;
    o->max_iter = 200;                         // cuda/nmf.cu:10
    o->iter_check = NMF_ITER_CHECK_DEFAULT;    // cuda/nmf.cu:9
    o->verbose = 0;
    o->path = NMF_PATH_AUTO;
    o->use_graph = NMF_GRAPH_AUTO;
    o->device = -1;
    o->stream = nullptr;
    o->comm = nullptr;
    o->nsplit_h = 0;
    o->nsplit_w = 0;
    o->fast_divide = 0;
    o->restart_lanes = 0;
    o->split_kernel = 0;
    o->n_devices = 0;
    o->devices = nullptr;
    o->emulate_shards = 0;
}

static double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ------------------------------------------------------------------------------ solver
struct nmf_solver {
    int M = 0, N = 0, K = 0;       // logical (local) dims
    int Mp = 0, Np = 0, Kp = 0;    // padded device dims
    int p1_trim = 0;               // 16-column kernels, Kc % 64 != 0: steps of product 1 skipped because they cover zero padding only: (Kc - pad4(K)) / 4
    int Kc = 0;                    // the K the chosen kernel computes on: a multiple of 16, <= Kp (FusedArgs::Kc / SplitArgs::Kc); the rest of Kp is zero padding
    int path = NMF_PATH_FUSED;
    int use_graph = 1;
    int nsplit_h = 1, nsplit_w = 1;
    int fast_divide = 0;
    // split path (nmf_split16_impl.h): four waves per 16 owned columns, normalisers summed in-stream, `batch` (W, H) pairs per launch
    bool split = false;
    int batch = 1;
    int cus = 256;                 // compute units the launch planning counted (the device's, plan_solver)
    int split_batch = 1;           // restarts in the whole update_div_restarts call this solver serves a share of (>= batch): what pick_split sees
    int ns_h = 1, ns_w = 1;        // workgroup-level splits of the reduction dimension on the split path
    int nw_h = 4, nw_w = 4;        // waves per workgroup of the two half-steps (8 where K = 64 and the reduction length allows)
    float *vpart = nullptr;        // [batch][ns][Kp] per-split sums of the streamed factor
    int *active_d = nullptr;       // [batch] device flags, nullptr = all pairs iterate
    int *active_own = nullptr;     // ... when allocated after creation (set_active on an unbatched solver), outside the arena
    bool x_shared = false;         // X belongs to another solver (update_div_restarts lanes)
    bool normW_fresh = false;      // normW = max(colsum(W), EPS) of the current W (left by the W-step's apply kernel): the H-step
                                   // may skip its column-sum launch.  Cleared by everything else that writes W.
    int x_in_range = 0;            // X verified at upload: every entry 0 or in [EPS, 2^60] (FusedArgs::x_in_range)
    unsigned *range_flag = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int stream_device = -1;        // device the pooled stream belongs to
    nmf_comm *comm = nullptr;      // sharded over N with in-library RCCL all-reduce
    bool external_reduce = false;  // sharded, caller reduces the partial buffer
    bool comm_warm = false;        // one eager all-reduce has run on this communicator (before any capture)
    // device state
    float *W = nullptr, *H = nullptr, *X = nullptr;
    float *normW = nullptr, *normH = nullptr, *rowpart = nullptr;
    float *partials = nullptr;     // split slabs
    float *vsum_part = nullptr;    // nsplit_w x Kp row sums of H per split, written by the W-step kernel (FusedArgs::vsum_part)
    float *psum = nullptr;         // Mp*Kp + Kp floats: [sum_g Z*H' ; rowsum(H)] (all-reduce operand)
    float *psum_owned = nullptr;   // the allocation behind psum unless the caller supplied one
    void *arena = nullptr;         // the one device allocation behind every buffer below (Arena)
    double *chk_part = nullptr, *chk_out = nullptr;
    std::vector<double> chk_host;  // 3 doubles per pair (pageable: a pinned allocation costs more than a lifetime of 24-byte copies)
    bool xc_valid = false;         // xc3 holds the X-only terms of the check for the X now resident (computed at the first check)
    double *xc_part = nullptr, *xc3 = nullptr, *sum64 = nullptr;   // the check's X-only terms and fp64 factor sums (launch_x_consts / launch_check_compose)
    int chk_groups = 0;
    int ns_chk = 1;                // 16-column kernel: cuts of the check's reduction over M (chk_groups = column groups x ns_chk)
    float *Z = nullptr, *WtZ = nullptr, *ZHt = nullptr;   // unfused temporaries (cuda/nmf.cu:94-96)
    float *staging = nullptr;      // unpadded upload/download staging (max of the three matrices)
    size_t staging_count = 0;
    // graph
    // captured iterations: one graph per batch size (kGraphIters); a replay costs 10-16 us of host / launch time whatever it
    // holds, which is a third of a small problem's iteration, so long runs replay 32 iterations at a time
    struct Level { hipGraph_t g = nullptr; hipGraphExec_t e = nullptr; bool ready = false, failed = false; };
    Level level[3];
    bool graph_ready = false;      // the single-iteration graph (level 2) exists: capture works for this solver
    bool timing_failed = false;    // a hipEvent of the piece timers could not be created: t[] would under-report
    // piece timing (eager, hipEvent pairs)
    bool timing = false;
    struct Ev { int which; hipEvent_t a, b; };
    std::vector<Ev> events;
    double t_setup = 0.0;
    double block_budget_s = 1e9;   // sharded runs: deadline of the next wait (solver_run keeps it at a multiple of what a block has taken so far)
};

// Every device buffer of a solver comes out of ONE allocation: a drop-in update_div call creates and destroys a solver, and a
// dozen hipMalloc / hipFree pairs were a millisecond of a 12 ms call at cfg2.  Buffers are registered while the shapes are
// worked out and carved (256-byte aligned) when the total is known.
struct Arena {
    struct Req { void **p; size_t bytes; bool zero; };
    std::vector<Req> reqs;
    void reserve(void **p, size_t bytes, bool zero = false) { *p = nullptr; if (bytes) reqs.push_back({p, bytes, zero}); }
    int commit(void **base_out, hipStream_t st) {
        size_t total = 0;
        for (auto &r : reqs) total += (r.bytes + 255) & ~(size_t)255;
        char *base = nullptr;
        const bool trace = getenv("NMF_RESTART_TRACE") != nullptr;
        const double t0 = now_s();
        if (const char *lim = getenv("NMF_FAULT_ARENA_LIMIT_MB")) {   // fault injection for the tests: a device with this much free memory
            if ((double)total > atof(lim) * 1048576.0) { set_err("a solver of %.1f MiB does not fit the device's free memory (NMF_FAULT_ARENA_LIMIT_MB)", total / 1048576.0); return NMF_ERR_NOMEM; }
        }
        {
            const hipError_t me = hipMalloc((void **)&base, total ? total : 256);
            if (me == hipErrorOutOfMemory) { (void)hipGetLastError(); set_err("a solver of %.1f GiB does not fit the device's free memory", total / 1073741824.0); return NMF_ERR_NOMEM; }
            HIPCHK(me);
        }
        const double t1 = now_s();
        *base_out = base;
        // one 16-byte-store kernel over the whole arena instead of a hipMemsetAsync per zero-initialised buffer (those took 18-21 ms
        // on the gold shape's buffers: nmf_kernels.h, launch_zero); buffers that need no zeroing are cleared too, at ~3 TB/s
        size_t off = 0, zeroed = total;
        for (auto &r : reqs) { *r.p = base + off; off += (r.bytes + 255) & ~(size_t)255; }
        HIPCHK(launch_zero(base, total, st));
        if (trace) {
            const double t2 = now_s();
            (void)hipStreamSynchronize(st);
            fprintf(stderr, "nmf arena: %.1f MiB: hipMalloc %.3f ms, %.1f MiB of memsets enqueued in %.3f ms, drained in %.3f ms\n", total / 1048576.0, (t1 - t0) * 1e3,
                    zeroed / 1048576.0, (t2 - t1) * 1e3, (now_s() - t2) * 1e3);
        }
        return NMF_OK;
    }
};

// Library-owned streams are pooled per device: creating one costs 1.5-2 ms (a hardware queue behind it), which is a fifth of a
// one-shot update_div on the reference's own problem.  A destroyed solver hands its (drained) stream back; the next solver on
// that device takes it.  At most kStreamPool idle streams per device are kept; the rest are destroyed.
namespace {
constexpr int kStreamPool = 8, kPoolDevices = 64;
std::mutex g_pool_mu;
std::vector<hipStream_t> g_pool[kPoolDevices];
int acquire_stream(hipStream_t *out) {
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev >= 0 && dev < kPoolDevices) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (!g_pool[dev].empty()) { *out = g_pool[dev].back(); g_pool[dev].pop_back(); return NMF_OK; }
    }
    HIPCHK(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    return NMF_OK;
}
void release_stream(hipStream_t st) {
    int dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kPoolDevices && hipStreamQuery(st) == hipSuccess) {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if ((int)g_pool[dev].size() < kStreamPool) { g_pool[dev].push_back(st); return; }
    }
    (void)hipGetLastError();
    (void)hipStreamDestroy(st);
}
}  // namespace

// launch_apply_w_colsum walks a column of W with one 1024-thread workgroup: fine up to 64 rows per thread
constexpr int kMaxRowsApplyColsum = 65536;

// The launch planning below (how a half-step's reduction is cut, which kernel family serves a shape) counts compute units.  The number comes
// from the device the solver is created on (hipDeviceAttributeMultiprocessorCount: 256 on a whole MI355X, fewer on a CPX / NPS partition of
// one); nmf_plan_describe, which touches no device, plans for 256.  NMF_PLAN_CUS overrides both (tests, A/B).  The fitted constants of the
// models (occupancy penalties, crossovers in elements per CU) were measured on 256 CUs and are carried over per CU.
constexpr int kDefaultCus = 256;
static int plan_cus_override() {
    const char *e = getenv("NMF_PLAN_CUS");
    const int v = e ? atoi(e) : 0;
    return (v >= 1 && v <= 4096) ? v : 0;
}
static int device_cus() {
    if (const int o = plan_cus_override()) return o;
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); return kDefaultCus; }
    return n;
}

// workgroups of the 64-column kernel a CU holds at once, by the registers and the LDS of its instantiations (profiles/r04_kernel_resources.md:
// K = 16: 90 registers; K <= 64: <= 120; K <= 128: <= 168 and 44 KiB; K <= 256: two by __launch_bounds__; above: one)
static int k16_resident(int kc) { return kc <= 64 ? 4 : (kc <= 128 ? 3 : (kc <= 256 ? 2 : 1)); }

// The rule of rounds 1-3 (and of the kernels without an occupancy table: the wave-pair and the 32-column kernels): workgroups per
// split-less launch = batch * ceil(Q/q_per_group); aim for >= 512 workgroups (2 per CU), keep >= 2 chunks of 32 per split.
static int pick_nsplit_512(int q_extent, int p_extent, int q_per_group, int batch, int cus) {
    const long nq = (long)((q_extent + q_per_group - 1) / q_per_group) * (batch > 1 ? batch : 1);
    if (nq >= cus) return 1;
    int ns = (int)((2L * cus + nq - 1) / nq);
    const int max_ns = (p_extent / 32) / 2 > 0 ? (p_extent / 32) / 2 : 1;
    if (ns > max_ns) ns = max_ns;
    if (ns > 64) ns = 64;
    if (ns < 1) ns = 1;
    return ns;
}

// How many ways the 64-column kernel cuts a half-step's reduction (each cut = one more workgroup per 64 owned columns and one more slab
// for the apply launch).  A small model of the launch, fitted to same-box sweeps (profiles/r04_nsplit_by_occupancy.log,
// r04_nsplit_model.log), picks the cheapest count:
//   * balance: nq * ns workgroups over 256 CUs -- ceil(wgs / 256) * 256 / wgs (313 row blocks unsplit keep half the chip waiting for
//     the CUs that got two: 0.519 ms against 0.315 with four cuts at 20000 x 4096 x 128);
//   * occupancy: a CU holding r = min(resident, ceil(wgs / 256)) workgroups runs at KT / (KT + a_r), a = 1.0 / 0.2 / 0.08 / 0 for r = 1 / 2 /
//     3 / >= 4 -- one wave per SIMD cannot keep the MFMA pipe fed, and the less so the shorter the chains (K = 16: 0.104 -> 0.056 ms for the
//     H-step of 4096 x 16384 from one to four workgroups per CU; K = 256: 0.504 -> 0.485);
//   * fixed work per workgroup: prologue + epilogue ~ two chunks of 32, so 1 + 2 ns / chunks;
//   * slabs: 0.4 % + 0.2 % per cut for writing them and for the apply launch that sums them.
// A launch that carries `batch` pairs (restarts) hands out batch times the workgroups and is planned as such -- with the batch of the
// whole update_div_restarts call, so that a restart gets the same cuts, hence the same bits, wherever it runs.
static int pick_nsplit(int q_extent, int p_extent, int q_per_group, int batch, int resident, int kt, int cus) {
    if (resident < 1 || kt < 1) return pick_nsplit_512(q_extent, p_extent, q_per_group, batch, cus);
    const long nq = (long)((q_extent + q_per_group - 1) / q_per_group) * (batch > 1 ? batch : 1);
    const int chunks = p_extent / 32 > 0 ? p_extent / 32 : 1;
    int max_ns = chunks / 2 > 0 ? chunks / 2 : 1;
    if (max_ns > 64) max_ns = 64;
    int best = 1;
    double best_cost = 0.0;
    for (int ns = 1; ns <= max_ns; ++ns) {
        const long wgs = nq * ns, per_cu = (wgs + cus - 1) / cus;
        const double balance = (double)(per_cu * cus) / (double)wgs;
        const long r = per_cu < resident ? per_cu : resident;
        const double a = r <= 1 ? 1.0 : (r == 2 ? 0.2 : (r == 3 ? 0.08 : 0.0));
        const double cost = balance * ((kt + a) / kt) * (1.0 + 2.0 * ns / chunks) * (ns > 1 ? 1.004 + 0.002 * ns : 1.0);
        if (ns == 1 || cost < best_cost - 1e-12) { best = ns; best_cost = cost; }
    }
    return best;
}

static int solver_init(nmf_solver *s, int M, int N, int K, const nmf_opts &o, const nmf_solver *x_from = nullptr, int batch = 1, int split_batch = 0);

extern "C" int nmf_solver_create(nmf_solver **out, int M, int N, int K, const nmf_opts *opts_in) {
    return nmf_solver_create_batched(out, M, N, K, 1, opts_in);
}
static int create_batched(nmf_solver **out, int M, int N, int K, int batch, const nmf_opts *opts_in, int split_batch);
extern "C" int nmf_solver_create_batched(nmf_solver **out, int M, int N, int K, int batch, const nmf_opts *opts_in) {
    return create_batched(out, M, N, K, batch, opts_in, 0);
}
static int create_batched(nmf_solver **out, int M, int N, int K, int batch, const nmf_opts *opts_in, int split_batch) {
    if (!out || M <= 0 || N <= 0 || K <= 0 || batch < 1 || batch > 65535) { set_err("nmf_solver_create: bad arguments"); return NMF_ERR_ARG; }
    nmf_opts o;
    if (opts_in) o = *opts_in; else nmf_default_opts(&o);
    if (batch > 1 && o.comm) { set_err("nmf_solver_create_batched: batched solvers do not shard"); return NMF_ERR_UNSUPPORTED; }
    if (o.device >= 0) HIPCHK(hipSetDevice(o.device));
    nmf_solver *s = new nmf_solver();
    const int st = solver_init(s, M, N, K, o, nullptr, batch, split_batch);
    if (st != NMF_OK) {   // release whatever was allocated before the failure
        nmf_solver_destroy(s);
        return st;
    }
    *out = s;
    return NMF_OK;
}

// Which kernel family serves a shape.  The split kernel (four waves per 16 owned columns, normalisers in-stream, no helper
// launches) wins wherever one workgroup per 64 owned columns leaves CUs idle or needs many partial slabs; the 64-column
// kernel wins once both half-steps fill the chip on their own (measured crossover: tools/shape_bench.py).
static int pick_split(int q_valid, int p_extent, int sc_rows, int batch, int wg_per_cu, int cus);
// how far a lone problem's split-kernel launch is from filling the chip evenly: CUs idle (fewer workgroups than CUs) or a part of
// them running a second round (more), as a factor on the launch's time
static double split_kernel_balance(int q, int p, int wg_per_cu, int cus) {
    const int qv = (q + 31) & ~31, tasks = qv / 16;
    const long wgs = (long)tasks * pick_split(qv, (p + 127) & ~127, 128, 1, wg_per_cu, cus);
    return wgs <= cus ? (double)cus / (double)wgs : (double)(((wgs + cus - 1) / cus) * cus) / (double)wgs;
}
static int split_pad_k(int K) { return K <= 256 ? pad32(K) : 0; }   // K in HBM: padded to 32 like the reference; the kernel computes on split_compute_k(K)
static bool want_split(int M, int N, int K, const nmf_opts &o, int cus) {
    const int kp = split_pad_k(K);
    if (!kp || !split_step_supports(kp) || o.split_kernel < 0 || o.path == NMF_PATH_UNFUSED) return false;
    if (((size_t)M + 127) * kp >= ((size_t)1 << 30) || ((size_t)N + 127) * kp >= ((size_t)1 << 30)) return false;
    if (o.split_kernel > 0) return true;
    // Measured crossover (tools/crossover.py: iteration time of both families).  Rounds 2-4 fitted it three times as the 64-column kernel
    // gained instantiations; the last fit (profiles/r04_crossover_model_splits.log) is against its model-chosen splits (pick_nsplit):
    //   K <= 16 (the split kernel computes on 32):  +16 % at 2^22 elements, level at 2^23, behind beyond
    //   K <= 112:  +8-27 % through 2^23 (K = 32: +11 %, 64: +8 %, 100: level), then level or 5-6 % behind at 2^24
    //   K <= 128:  +12-23 % at 2^22, level at 2^23
    //   K <= 256:  ahead through 2^21 only (K = 200: 8-11 % behind at 2^21.8 .. 2^23; K = 256: +47 % at 2^20, +3 % at 2^21, mixed at 2^22)
    // and, whatever the K, behind by 17-29 % from 2^22 elements on where its own workgroup count does not fit the chip -- 16 owned columns
    // per workgroup and at most 256 workgroups: 157 or 314 of them (N = 2500, 5000) leave a third of the CUs idle or waiting for a second
    // round, where the 64-column kernel cuts its reduction to fit (3000 x 5000 x 100: 198 us against 140).
    const int kc = split_compute_k(K);
    // (the fits are of a 256-CU chip and carried over per CU: `unit` = 2^21 elements on 256 CUs, 2^13 per CU)
    const size_t mn = (size_t)M * (size_t)N, unit = (((size_t)1 << 21) * (size_t)cus) / 256;
    const size_t lim = K <= 16 ? 4 * unit : (kc <= 112 ? 6 * unit : (kc <= 128 ? 4 * unit : unit));
    if (mn > lim) return false;
    if (mn >= 2 * unit) {
        const int wpc = kp <= 64 ? 2 : 1;
        const double bh = split_kernel_balance(N, M, wpc, cus), bw = split_kernel_balance(M, N, wpc, cus);
        if ((bh > bw ? bh : bw) >= 1.3) return false;
    }
    return true;
}
// workgroup-level split count of the reduction dimension for Q/16 column groups over `nsc` superchunks of 128.
// A single pair: one workgroup per CU is the target, never more workgroups than CUs (a second round costs more than the
// shorter loop saves: 264 workgroups on the gold shape's H-step took 20.4 us against 15.5 us for 176).
// A launch that carries `batch` pairs (restarts) already hands out batch x column-groups workgroups, and every further split
// only multiplies fixed costs (prologue, epilogue, slabs, the apply launch): the split shrinks with the batch and stops at
// `wg_per_cu` workgroups per CU -- two where the LDS image allows it (K <= 64; K = 128 with a batch: one image), because a
// co-resident workgroup fills the other's quotient / barrier / first-touch time.  Measured (tools/restart_split_sweep.py,
// profiles/r03_restart_sweep.log), 16 restarts of 4096 x 350 x 128, H-step split 1 / 2 / 4 / 11: 295 / 263 / 266 / 275 us per
// iteration of the batch; 16 x cfg2, W-step split 1 / 2 / 4: 319 / 323 / 332.  Splits that divide the superchunk count evenly
// are preferred (32 superchunks three ways -- 11, 11, 10 -- cost 99 us where four ways cost 88).  `batch` is the restart count
// of the whole update_div_restarts call (not of one device's or one chunk's share), so that a restart gets the same split --
// hence the same bits -- wherever it runs.
static int pick_split(int q_valid, int p_extent, int sc_rows, int batch, int wg_per_cu, int cus) {
    const int tasks = q_valid / 16, nsc = p_extent / sc_rows;
    if (nsc <= 1) return 1;
    int S;
    if (batch <= 1) {
        if (4 * tasks >= 3 * cus) return 1;   // three quarters of the CUs busy: a second slab costs more than the idle quarter
        S = cus / tasks;
    } else {
        const long groups = (long)tasks * batch, target = (long)cus * (wg_per_cu > 1 ? 2 : 1);
        if (groups >= target) return 1;
        S = (int)((target + groups - 1) / groups);
        for (int d = S; d <= 2 * S && d <= nsc; ++d) if (nsc % d == 0) { S = d; break; }   // an even division if one is near
    }
    if (S < 1) S = 1;
    if (S > nsc) S = nsc;
    const int scps = (nsc + S - 1) / S;
    return (nsc + scps - 1) / scps;
}

// Everything about a solver that follows from the shape and the options alone -- kernel family, padded dims, split counts,
// waves per workgroup: no HIP call in here, so the decision tables can be tested on a machine without a GPU (nmf_plan_describe)
static int plan_solver(nmf_solver *s, int M, int N, int K, const nmf_opts &o, int batch, int split_batch, int cus) {
    s->M = M; s->N = N; s->K = K;
    int path = o.path;
    s->batch = batch;
    s->cus = cus;
    s->split = want_split(M, N, K, o, cus);
    // the split kernel streams whole superchunks of 128: zero padding is invariant under the updates and adds nothing to any sum
    s->Mp = s->split ? ((M + 127) & ~127) : pad32(M);
    s->Np = s->split ? ((N + 127) & ~127) : pad32(N);
    if (s->split) path = NMF_PATH_FUSED;
    // the 16x16x4 kernels address the streamed factor with 32-bit lane offsets and have no 64-bit fallback
    const bool k16_too_tall = fused_pad_k(K) >= 32 && (size_t)fused_pad_k(K) * (size_t)s->Mp >= ((size_t)1 << (fused_pad_k(K) > kMaxK16 ? 30 : 31));
    if (path == NMF_PATH_AUTO) path = (fused_pad_k(K) && !k16_too_tall) ? NMF_PATH_FUSED : NMF_PATH_UNFUSED;
    if (path == NMF_PATH_FUSED && k16_too_tall) { set_err("fused path supports M*K < 2^31"); return NMF_ERR_UNSUPPORTED; }
    if (path == NMF_PATH_FUSED) {
        if (!fused_pad_k(K)) { set_err("fused path supports K <= %d", kMaxFusedK); return NMF_ERR_UNSUPPORTED; }
        // K in HBM is padded to 32 (cuda/matrix.cuh:7) and the 16-column kernels compute on the next multiple of 16
        s->Kp = s->split ? split_pad_k(K) : fused_pad_k(K);
        s->Kc = s->split ? split_compute_k(K) : fused_compute_k(K);
    } else {
        s->Kp = pad32(K);
        s->Kc = s->Kp;
    }
    s->path = path;
    // product 1 of the 64-column kernel steps through K four at a time: where the last two or three steps hold zero padding only
    // (K = 100 on the K = 112 kernel; K = 200 on K = 208) a variant whose chain ends that many steps early runs (nmf_fused16_impl.h: TRIM)
    s->p1_trim = 0;
    if (path == NMF_PATH_FUSED && !s->split && s->Kc >= 16 && s->Kc <= 256 && !getenv("NMF_NO_P1_TRIM")) {
        const int zero_steps = (s->Kc - ((K + 3) & ~3)) / 4;
        s->p1_trim = zero_steps >= 3 ? 3 : (zero_steps == 2 ? 2 : 0);
        if (s->Kc == 16 && s->p1_trim > 2) s->p1_trim = 2;   // four steps in all: two of them stay
    }
    // a batch of (W, H) pairs per launch: the split kernel, or the 64-column kernel where its W-step delivers the row sums of H
    // (blockIdx.y = pair: nmf_fused16_impl.h); the 32-column and the wave-pair kernels and the operator path take one pair
    if (batch > 1 && !s->split && !(path == NMF_PATH_FUSED && fused_takes_batch(s->Kp) && s->Mp <= kMaxRowsApplyColsum)) {
        set_err("batched solvers need the split kernel or the 64-column kernel (K <= 576, M <= 65536)");
        return NMF_ERR_UNSUPPORTED;
    }
    s->use_graph = o.use_graph > 0 ? 1 : 0;   // 0: eager + per-piece timers in update_div_ex; < 0: eager, untimed
    s->fast_divide = o.fast_divide;
    if (s->split) {
        // eight waves per workgroup (two per SIMD from a single workgroup per CU) where K = 64 leaves the LDS for eight sub-images
        // (measured: 3 % on cfg2 alone, but 17 % slower than two independent four-wave workgroups per CU once a batch fills
        //  the chip -- the eight waves share one barrier and run in phase -- so it stays an experiment: NMF_SPLIT_NW=8)
        const char *nwe = getenv("NMF_SPLIT_NW");
        const bool nw8 = s->Kc == 64 && nwe && nwe[0] == '8';
        s->nw_h = (nw8 && s->Mp % 256 == 0) ? 8 : 4;
        s->nw_w = (nw8 && s->Np % 256 == 0) ? 8 : 4;
        const int sb = split_batch > 0 ? split_batch : batch;   // restarts in the whole call (pick_split)
        s->split_batch = sb;
        const int wpc = (s->Kp <= 64 || (s->Kp <= 128 && sb > 1)) ? 2 : 1;   // workgroups the LDS image lets share a CU (split_args: single_image)
        s->ns_h = o.nsplit_h > 0 ? o.nsplit_h : pick_split((N + 31) & ~31, s->Mp, 32 * s->nw_h, sb, wpc, cus);
        s->ns_w = o.nsplit_w > 0 ? o.nsplit_w : pick_split((M + 31) & ~31, s->Np, 32 * s->nw_w, sb, wpc, cus);
        const int nsc_h = s->Mp / (32 * s->nw_h), nsc_w = s->Np / (32 * s->nw_w);
        if (s->ns_h > nsc_h) s->ns_h = nsc_h;
        if (s->ns_w > nsc_w) s->ns_w = nsc_w;
        s->nsplit_h = s->ns_h; s->nsplit_w = s->ns_w;
        s->chk_groups = check_num_groups(s->Np, s->Kp);
    } else if (path == NMF_PATH_FUSED) {
        const int qg = fused_cols_per_group(s->Kp);
        const int sb = split_batch > 0 ? split_batch : batch;   // restarts in the whole call: a restart gets the same splits wherever it runs
        s->split_batch = sb;
        const char *rese = getenv("NMF_NSPLIT_RESIDENT");   // A/B: 0 = the 512-workgroup rule of rounds 1-3 whatever the K
        // the wave-pair kernel (K > 576) holds one workgroup per CU: the same model with r = 1 (balance and fixed work decide; 3000 x 20000 x 700
        // 3.69 -> 2.99 ms per iteration, 3000 x 3000 x 1000 0.88 -> 0.71: profiles/r04_nsplit_model.log)
        const int res = rese ? atoi(rese) : (s->Kp > kMaxK16 ? 1 : (!getenv("NMF_FUSED_VARIANT") ? k16_resident(s->Kc) : 0));
        s->nsplit_h = o.nsplit_h > 0 ? o.nsplit_h : pick_nsplit(s->Np, s->Mp, qg, sb, res, s->Kc / 16, cus);
        s->nsplit_w = o.nsplit_w > 0 ? o.nsplit_w : pick_nsplit(s->Mp, s->Np, qg, sb, res, s->Kc / 16, cus);
        if (s->nsplit_h > s->Mp / 32) s->nsplit_h = s->Mp / 32;
        if (s->nsplit_w > s->Np / 32) s->nsplit_w = s->Np / 32;
        s->chk_groups = check_num_groups(s->Np, s->Kp);
    } else {
        s->nsplit_w = 16;                                  // split-K slabs for the Z*H' GEMM
        s->chk_groups = reduce_num_groups((size_t)s->Mp * s->Np);
    }
    if (path == NMF_PATH_FUSED && fused_takes_batch(s->Kp)) {
        // The KL check is an H-step-shaped launch (product 1 only) on the 64-column kernel whichever family iterates: six workgroups for the
        // reference's 350 columns -- 183 us, six iterations' worth, +30 % on a run that checks every 25 iterations.  Its reduction over M is cut
        // by the same model as a half-step's, always as for a lone problem: a restart's KL has the same bits alone and in a batch.
        s->ns_chk = pick_nsplit(s->Np, s->Mp, 64, 1, k16_resident(s->Kc), s->Kc / 16, cus);
        if (s->ns_chk > s->Mp / 32) s->ns_chk = s->Mp / 32 > 0 ? s->Mp / 32 : 1;
        s->chk_groups = check_num_groups(s->Np, s->Kp) * s->ns_chk;
    }
    return NMF_OK;
}

// the planning of nmf_solver_create_batched as one line (nmf_solver_describe's), without creating anything: tests of the
// dispatch tables on machines without a GPU, and "what would run" queries
extern "C" int nmf_plan_describe(int M, int N, int K, int batch, const nmf_opts *opts_in, char *buf, int buflen) {
    if (M <= 0 || N <= 0 || K <= 0 || batch < 1 || !buf || buflen <= 0) return NMF_ERR_ARG;
    nmf_opts o;
    if (opts_in) o = *opts_in; else nmf_default_opts(&o);
    nmf_solver s;
    const int cus = plan_cus_override() ? plan_cus_override() : kDefaultCus;   // no device is touched here
    NMFCHK(plan_solver(&s, M, N, K, o, batch, 0, cus));
    return nmf_solver_describe(&s, buf, buflen);
}

static int solver_init(nmf_solver *s, int M, int N, int K, const nmf_opts &o, const nmf_solver *x_from, int batch, int split_batch) {
    const double t0 = now_s();
    NMFCHK(plan_solver(s, M, N, K, o, batch, split_batch, device_cus()));
    const int path = s->path;
    s->comm = (nmf_comm *)o.comm;
    if (s->comm) s->block_budget_s = nmf_comm_timeout_s();   // waits outside solver_run (a check after iterate()) carry the communicator's deadline too
    if (o.stream) { s->stream = (hipStream_t)o.stream; s->own_stream = false; }
    else { NMFCHK(acquire_stream(&s->stream)); s->own_stream = true; s->stream_device = -1; (void)hipGetDevice(&s->stream_device); }

    const size_t mk = (size_t)s->Mp * s->Kp, kn = (size_t)s->Kp * s->Np, mn = (size_t)s->Mp * s->Np;
    Arena ar;
    bool zero_slabs = false;
    ar.reserve((void **)&s->W, mk * batch * sizeof(float), true);
    ar.reserve((void **)&s->H, kn * batch * sizeof(float), true);
    if (x_from) { s->X = x_from->X; s->x_shared = true; s->x_in_range = x_from->x_in_range; }   // read-only, already uploaded
    else ar.reserve((void **)&s->X, mn * sizeof(float), true);
    ar.reserve((void **)&s->normW, ((size_t)s->Kp * batch) * sizeof(float));
    ar.reserve((void **)&s->normH, ((size_t)s->Kp * batch) * sizeof(float));
    ar.reserve((void **)&s->rowpart, ((size_t)row_sum_blocks(s->Np) * s->Kp * (s->split ? 1 : batch)) * sizeof(float));
    ar.reserve((void **)&s->psum_owned, (mk + (size_t)s->Kp) * sizeof(float));
    if (s->split) {
        size_t pc = (s->ns_w > 1 || batch == 1) ? (size_t)s->ns_w * mk : 0;   // an unbatched solver's W-step may always need slabs (sharded runs)
        if (s->ns_h > 1 && (size_t)s->ns_h * kn > pc) pc = (size_t)s->ns_h * kn;
        ar.reserve((void **)&s->partials, (pc * batch) * sizeof(float));
        zero_slabs = true;   // rows / columns of pure zero padding get no workgroup: their slab entries are never written and must read as zero
        const int nsm = s->ns_h > s->ns_w ? s->ns_h : s->ns_w;
        ar.reserve((void **)&s->vpart, ((size_t)nsm * s->Kp * batch) * sizeof(float));
    } else if (path == NMF_PATH_FUSED) {
        size_t pc = 0;
        if (s->nsplit_h > 1) pc = (size_t)s->nsplit_h * kn;
        if ((size_t)s->nsplit_w * mk > pc) pc = (size_t)s->nsplit_w * mk;   // W-step may always need slabs (sharded; a batch: always)
        ar.reserve((void **)&s->partials, (pc * batch) * sizeof(float));
        if ((s->nsplit_w > 1 || s->split_batch > 1) && fused_streams_vsum(s->Mp, s->Kp)) ar.reserve((void **)&s->vsum_part, ((size_t)s->nsplit_w * s->Kp * batch) * sizeof(float));
    } else {
        ar.reserve((void **)&s->Z, (mn) * sizeof(float));
        ar.reserve((void **)&s->WtZ, (kn) * sizeof(float));
        ar.reserve((void **)&s->ZHt, (mk) * sizeof(float));
        ar.reserve((void **)&s->partials, ((size_t)s->nsplit_w * mk) * sizeof(float));
    }
    ar.reserve((void **)&s->chk_part, sizeof(double) * 3 * (size_t)s->chk_groups * (size_t)batch);   // one set per pair: check_all evaluates every pair in one launch
    ar.reserve((void **)&s->chk_out, sizeof(double) * 3 * (size_t)batch);
    s->chk_host.assign(3 * (size_t)batch, 0.0);
    if (batch > 1) ar.reserve((void **)&s->active_d, sizeof(int) * (size_t)batch);   // from the start: captured graphs never hold a stale null pointer
    ar.reserve((void **)&s->range_flag, sizeof(unsigned));
    if (path == NMF_PATH_FUSED) {
        ar.reserve((void **)&s->xc_part, sizeof(double) * 3 * (size_t)kXConstGroups);
        ar.reserve((void **)&s->xc3, sizeof(double) * 3, true);
        ar.reserve((void **)&s->sum64, sizeof(double) * ((size_t)s->Kp + kSum64Blocks));
    }
    s->staging_count = (size_t)M * N;
    if ((size_t)M * K > s->staging_count) s->staging_count = (size_t)M * K;
    if ((size_t)K * N > s->staging_count) s->staging_count = (size_t)K * N;
    // a batched solver moves all its pairs through staging behind ONE synchronisation (upload_pairs / download_pairs)
    if (batch > 1 && (size_t)batch * ((size_t)M * K + (size_t)K * N) > s->staging_count) s->staging_count = (size_t)batch * ((size_t)M * K + (size_t)K * N);
    ar.reserve((void **)&s->staging, s->staging_count * sizeof(float));
    for (auto &r : ar.reqs) if (zero_slabs && (r.p == (void **)&s->partials || r.p == (void **)&s->psum_owned)) r.zero = true;
    NMFCHK(ar.commit(&s->arena, s->stream));
    s->psum = s->psum_owned;
    if (x_from && s->xc3) { HIPCHK(hipMemcpyAsync(s->xc3, x_from->xc3, sizeof(double) * 3, hipMemcpyDeviceToDevice, s->stream)); s->xc_valid = x_from->xc_valid; }
    if (batch > 1) {
        std::vector<int> ones((size_t)batch, 1);
        HIPCHK(hipMemcpyAsync(s->active_d, ones.data(), sizeof(int) * (size_t)batch, hipMemcpyHostToDevice, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    s->t_setup = now_s() - t0;
    return NMF_OK;
}

static void drop_graphs(nmf_solver *s) {
    for (auto &l : s->level) {
        if (l.e) (void)hipGraphExecDestroy(l.e);
        if (l.g) (void)hipGraphDestroy(l.g);
        l = nmf_solver::Level();
    }
    s->graph_ready = false;
}

extern "C" void nmf_solver_destroy(nmf_solver *s) {
    if (!s) return;
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    drop_graphs(s);
    for (auto &e : s->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    if (s->arena) (void)hipFree(s->arena);
    if (s->active_own) (void)hipFree(s->active_own);
    if (s->own_stream && s->stream) {   // back to the pool of its device (drained above)
        int cur = -1;
        (void)hipGetDevice(&cur);
        if (s->stream_device >= 0 && cur != s->stream_device) (void)hipSetDevice(s->stream_device);
        release_stream(s->stream);
        if (s->stream_device >= 0 && cur >= 0 && cur != s->stream_device) (void)hipSetDevice(cur);
    }
    delete s;
}

extern "C" int nmf_solver_path(const nmf_solver *s) { return s ? s->path : 0; }
// one line for logs and bench records: which kernel family runs, on which padded shape, with which split counts
extern "C" int nmf_solver_describe(const nmf_solver *s, char *buf, int buflen) {
    if (!s || !buf || buflen <= 0) return NMF_ERR_ARG;
    if (s->path != NMF_PATH_FUSED) snprintf(buf, (size_t)buflen, "unfused operators (gemm_kernel), Mp=%d Np=%d Kp=%d", s->Mp, s->Np, s->Kp);
    else if (s->split) snprintf(buf, (size_t)buflen, "split_step_kernel_k16<KT=%d> Mp=%d Np=%d Kp=%d splits(h,w)=(%d,%d) batch=%d", s->Kc / 16, s->Mp, s->Np, s->Kp, s->ns_h, s->ns_w, s->batch);
    else if (s->Kp > kMaxK16) snprintf(buf, (size_t)buflen, "fused_step_kernel_pair<KTH=%d> Mp=%d Np=%d Kp=%d nsplit(h,w)=(%d,%d)", s->Kc / 32, s->Mp, s->Np, s->Kp, s->nsplit_h, s->nsplit_w);
    else if (!getenv("NMF_FUSED_VARIANT")) snprintf(buf, (size_t)buflen, "fused_step_kernel_k16<KT=%d> Mp=%d Np=%d Kp=%d nsplit(h,w)=(%d,%d)%s", s->Kc / 16, s->Mp, s->Np, s->Kp, s->nsplit_h, s->nsplit_w,
                                                               s->p1_trim == 3 ? " p1_trim=3" : (s->p1_trim == 2 ? " p1_trim=2" : ""));
    else snprintf(buf, (size_t)buflen, "fused_step_kernel_v3<KT=%d> Mp=%d Np=%d Kp=%d nsplit(h,w)=(%d,%d)", s->Kp / 32, s->Mp, s->Np, s->Kp, s->nsplit_h, s->nsplit_w);
    if (s->cus != kDefaultCus) {   // planned for a chip (partition) with another CU count: say so
        const size_t n = strlen(buf);
        if (n + 12 < (size_t)buflen) snprintf(buf + n, (size_t)buflen - n, " cus=%d", s->cus);
    }
    return NMF_OK;
}
extern "C" void *nmf_solver_stream(nmf_solver *s) { return s ? (void *)s->stream : nullptr; }
extern "C" int nmf_solver_sync(nmf_solver *s) {
    if (!s) return NMF_ERR_ARG;
    HIPCHK(hipStreamSynchronize(s->stream));
    return NMF_OK;
}

// upload one matrix: (host|device) unpadded -> padded + clamp (read_matrix, cuda/nmf.cu:208-211)
static int upload_one(nmf_solver *s, float *dst, int rows_p, int cols_p, const float *src, int rows, int cols, bool host) {
    if (!src) return NMF_OK;
    const float *d = src;
    if (host) {
        HIPCHK(hipMemcpyAsync(s->staging, src, (size_t)rows * cols * sizeof(float), hipMemcpyHostToDevice, s->stream));
        d = s->staging;
    }
    const bool is_x = dst == s->X;
    if (dst == s->W) s->normW_fresh = false;
    if (is_x) HIPCHK(hipMemsetAsync(s->range_flag, 0, sizeof(unsigned), s->stream));
    HIPCHK(launch_pad_copy(dst, rows_p, cols_p, d, rows, cols, /*clamp=*/true, is_x ? s->range_flag : nullptr, s->stream));
    if (is_x) s->xc_valid = false;   // the X-only terms of the check are summed when a check first asks for them
    if (is_x) {
        unsigned flag = 1;
        HIPCHK(hipMemcpyAsync(&flag, s->range_flag, sizeof(unsigned), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
        s->x_in_range = flag == 0;
    } else if (host) {
        HIPCHK(hipStreamSynchronize(s->stream));   // staging is reused by the next matrix
    }
    return NMF_OK;
}

extern "C" int nmf_solver_upload(nmf_solver *s, const float *W, const float *H, const float *X) {
    if (!s) return NMF_ERR_ARG;
    NMFCHK(upload_one(s, s->W, s->Mp, s->Kp, W, s->M, s->K, true));
    NMFCHK(upload_one(s, s->H, s->Kp, s->Np, H, s->K, s->N, true));
    NMFCHK(upload_one(s, s->X, s->Mp, s->Np, X, s->M, s->N, true));
    return NMF_OK;
}
extern "C" int nmf_solver_upload_device(nmf_solver *s, const float *W, const float *H, const float *X) {
    if (!s) return NMF_ERR_ARG;
    NMFCHK(upload_one(s, s->W, s->Mp, s->Kp, W, s->M, s->K, false));
    NMFCHK(upload_one(s, s->H, s->Kp, s->Np, H, s->K, s->N, false));
    NMFCHK(upload_one(s, s->X, s->Mp, s->Np, X, s->M, s->N, false));
    return NMF_OK;
}

// write_matrix strips the padding (cuda/nmf.cu:228-231)
extern "C" int nmf_solver_download(nmf_solver *s, float *W, float *H) {
    if (!s) return NMF_ERR_ARG;
    if (W) {
        HIPCHK(launch_unpad_copy(s->staging, s->M, s->K, s->W, s->Mp, s->stream));
        HIPCHK(hipMemcpyAsync(W, s->staging, (size_t)s->M * s->K * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    if (H) {
        HIPCHK(launch_unpad_copy(s->staging, s->K, s->N, s->H, s->Kp, s->stream));
        HIPCHK(hipMemcpyAsync(H, s->staging, (size_t)s->K * s->N * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return NMF_OK;
}

extern "C" int nmf_solver_batch(const nmf_solver *s) { return s ? s->batch : 0; }
extern "C" int nmf_solver_uses_split_kernel(const nmf_solver *s) { return (s && s->split) ? 1 : 0; }
extern "C" int nmf_solver_upload_pair(nmf_solver *s, int b, const float *W, const float *H) {
    if (!s || b < 0 || b >= s->batch) return NMF_ERR_ARG;
    s->normW_fresh = false;
    NMFCHK(upload_one(s, s->W + (size_t)b * s->Mp * s->Kp, s->Mp, s->Kp, W, s->M, s->K, true));
    NMFCHK(upload_one(s, s->H + (size_t)b * s->Kp * s->Np, s->Kp, s->Np, H, s->K, s->N, true));
    return NMF_OK;
}
extern "C" int nmf_solver_download_pair(nmf_solver *s, int b, float *W, float *H) {
    if (!s || b < 0 || b >= s->batch) return NMF_ERR_ARG;
    if (W) {
        HIPCHK(launch_unpad_copy(s->staging, s->M, s->K, s->W + (size_t)b * s->Mp * s->Kp, s->Mp, s->stream));
        HIPCHK(hipMemcpyAsync(W, s->staging, (size_t)s->M * s->K * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    if (H) {
        HIPCHK(launch_unpad_copy(s->staging, s->K, s->N, s->H + (size_t)b * s->Kp * s->Np, s->Kp, s->stream));
        HIPCHK(hipMemcpyAsync(H, s->staging, (size_t)s->K * s->N * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    return NMF_OK;
}
// flags == nullptr re-activates every pair
extern "C" int nmf_solver_set_active(nmf_solver *s, const int *flags) {
    if (!s) return NMF_ERR_ARG;
    if (!flags && !s->active_d) return NMF_OK;
    if (!s->active_d) {   // batch == 1: a flag for the one pair; the graphs captured so far hold a null flag pointer
        drop_graphs(s);
        HIPCHK(hipMalloc((void **)&s->active_own, sizeof(int) * (size_t)s->batch));
        s->active_d = s->active_own;
    }
    std::vector<int> f((size_t)s->batch, 1);
    if (flags) for (int b = 0; b < s->batch; ++b) f[(size_t)b] = flags[b] ? 1 : 0;
    HIPCHK(hipMemcpyAsync(s->active_d, f.data(), sizeof(int) * (size_t)s->batch, hipMemcpyHostToDevice, s->stream));
    HIPCHK(hipStreamSynchronize(s->stream));
    return NMF_OK;
}

// All n pairs of a batched solver through the staging buffer (sized for it by solver_init) behind one synchronisation: a
// per-pair upload / download costs two stream synchronisations each, which is a third of a 16-restart call on a small problem
static int upload_pairs(nmf_solver *s, const matrix *W, const matrix *H, int n) {
    if (!s || n < 1 || n > s->batch) return NMF_ERR_ARG;
    const size_t mk = (size_t)s->M * s->K, kn = (size_t)s->K * s->N;
    if (s->staging_count < (size_t)n * (mk + kn)) {
        for (int b = 0; b < n; ++b) NMFCHK(nmf_solver_upload_pair(s, b, W[b].mat, H[b].mat));
        return NMF_OK;
    }
    s->normW_fresh = false;
    for (int b = 0; b < n; ++b) {
        float *sw = s->staging + (size_t)b * (mk + kn), *sh = sw + mk;
        HIPCHK(hipMemcpyAsync(sw, W[b].mat, mk * sizeof(float), hipMemcpyHostToDevice, s->stream));
        HIPCHK(launch_pad_copy(s->W + (size_t)b * s->Mp * s->Kp, s->Mp, s->Kp, sw, s->M, s->K, true, nullptr, s->stream));
        HIPCHK(hipMemcpyAsync(sh, H[b].mat, kn * sizeof(float), hipMemcpyHostToDevice, s->stream));
        HIPCHK(launch_pad_copy(s->H + (size_t)b * s->Kp * s->Np, s->Kp, s->Np, sh, s->K, s->N, true, nullptr, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return NMF_OK;
}
static int download_pairs(nmf_solver *s, const matrix *W, const matrix *H, int n) {
    if (!s || n < 1 || n > s->batch) return NMF_ERR_ARG;
    const size_t mk = (size_t)s->M * s->K, kn = (size_t)s->K * s->N;
    if (s->staging_count < (size_t)n * (mk + kn)) {
        for (int b = 0; b < n; ++b) NMFCHK(nmf_solver_download_pair(s, b, W[b].mat, H[b].mat));
        return NMF_OK;
    }
    for (int b = 0; b < n; ++b) {
        float *sw = s->staging + (size_t)b * (mk + kn), *sh = sw + mk;
        HIPCHK(launch_unpad_copy(sw, s->M, s->K, s->W + (size_t)b * s->Mp * s->Kp, s->Mp, s->stream));
        HIPCHK(launch_unpad_copy(sh, s->K, s->N, s->H + (size_t)b * s->Kp * s->Np, s->Kp, s->stream));
    }
    for (int b = 0; b < n; ++b) {
        float *sw = s->staging + (size_t)b * (mk + kn), *sh = sw + mk;
        HIPCHK(hipMemcpyAsync(W[b].mat, sw, mk * sizeof(float), hipMemcpyDeviceToHost, s->stream));
        HIPCHK(hipMemcpyAsync(H[b].mat, sh, kn * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    }
    HIPCHK(hipStreamSynchronize(s->stream));
    return NMF_OK;
}

// --------------------------------------------------------------------- tracing: roctx ranges
// One range per piece of an iteration (H-step, W-step, sums, apply, check, all-reduce) around its enqueue, so that
// `rocprofv3 --marker-trace --kernel-trace` attributes kernels to pieces.  The roctx library is dlopen()ed, and only when the
// process runs under a rocprofiler tool (LD_PRELOAD) or NMF_ROCTX=1 asks for it; NMF_ROCTX=0 switches it off.  Ranges bracket
// host-side enqueues: a replayed hipGraph has none per piece, so trace with eager launches (`nmf --timers`, use_graph = 0).
namespace {
struct Roctx {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        const char *e = getenv("NMF_ROCTX"), *pre = getenv("LD_PRELOAD");
        const bool want = e ? (e[0] == '1') : (pre && strstr(pre, "rocprofiler") != nullptr);
        if (!want) return;
        for (const char *n : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
            if (void *h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) {
                push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
                pop = (int (*)())dlsym(h, "roctxRangePop");
                if (push && pop) return;
                push = nullptr; pop = nullptr;
            }
        }
    }
};
Roctx &roctx() { static Roctx r; return r; }
const char *const kPieceName[10] = {"nmf:total", "nmf:h2d", "nmf:h_step", "nmf:w_step", "nmf:sums", "nmf:apply", "nmf:check", "nmf:allreduce", "nmf:d2h", "nmf:setup"};
}  // namespace

// --------------------------------------------------------------------- piece timing
struct PieceScope {
    nmf_solver *s; int idx = -1; bool marked = false;
    PieceScope(nmf_solver *s_, int which) : s(s_) {
        if (roctx().push && which >= 0 && which < 10) { roctx().push(kPieceName[which]); marked = true; }
        if (!s->timing) return;
        nmf_solver::Ev e; e.which = which; e.a = nullptr; e.b = nullptr;
        if (hipEventCreate(&e.a) != hipSuccess || hipEventCreate(&e.b) != hipSuccess) {
            if (e.a) (void)hipEventDestroy(e.a);
            s->timing_failed = true;      // reported by collect_timing
            return;
        }
        (void)hipEventRecord(e.a, s->stream);
        s->events.push_back(e);
        idx = (int)s->events.size() - 1;
    }
    ~PieceScope() {
        if (idx >= 0) (void)hipEventRecord(s->events[idx].b, s->stream);
        if (marked) roctx().pop();
    }
};

static int collect_timing(nmf_solver *s, double t[10]) {
    HIPCHK(hipStreamSynchronize(s->stream));
    for (auto &e : s->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess && t) t[e.which] += (double)ms * 1e-3;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    s->events.clear();
    if (s->timing_failed) {
        s->timing_failed = false;
        set_err("piece timers: hipEventCreate failed, t[] is incomplete");
        return NMF_ERR_HIP;
    }
    return NMF_OK;
}

// Sharded solvers never block without a deadline: a peer that has died or stalled leaves this rank's all-reduce kernel
// spinning.  `budget_s` is what the queued work may legitimately take.
static int wait_stream(nmf_solver *s, double budget_s, const char *what) {
    if (!s->comm) { HIPCHK(hipStreamSynchronize(s->stream)); return NMF_OK; }
    const int st = nmf_comm_wait(s->comm, s->stream, budget_s, what);
    if (st != NMF_OK) set_err("sharded run: %s did not complete (communicator aborted)", what);
    return st;
}

// --------------------------------------------------------------------- half-steps
static FusedArgs fused_args(nmf_solver *s) {
    FusedArgs a;
    a.W = s->W; a.H = s->H; a.X = s->X;
    a.U_out = nullptr; a.partials = s->partials; a.norm = nullptr;
    a.Mp = s->Mp; a.Np = s->Np; a.Kp = s->Kp; a.Kc = s->Kc; a.p1_trim = s->p1_trim; a.nsplit = 1; a.partial = 0; a.fast_divide = s->fast_divide > 0;
    a.x_in_range = s->fast_divide < 0 ? 0 : s->x_in_range;
    a.batch = s->batch; a.strideW = (size_t)s->Mp * s->Kp; a.strideH = (size_t)s->Kp * s->Np; a.active = s->active_d;
    return a;
}

static SplitArgs split_args(nmf_solver *s) {
    SplitArgs a;
    a.W = s->W; a.H = s->H; a.X = s->X;
    a.U_out = nullptr; a.partials = s->partials; a.vpart = s->vpart;
    a.Mp = s->Mp; a.Np = s->Np; a.Kp = s->Kp; a.Kc = s->Kc; a.nsplit = 1; a.batch = s->batch; a.force_partial = 0;
    a.nw_h = s->nw_h; a.nw_w = s->nw_w;
    // 64 < K <= 128: two workgroups per CU pay once a launch hands out more than one workgroup per CU (a batch of restarts)
    const bool mid_k = s->Kp > 64 && s->Kp <= 128;
    a.single_image = (mid_k && s->split_batch > 1 && !getenv("NMF_SPLIT_DOUBLE")) || (mid_k && getenv("NMF_SPLIT_SINGLE") != nullptr);
    a.Mv = (s->M + 31) & ~31; a.Nv = (s->N + 31) & ~31;
    a.strideW = (size_t)s->Mp * s->Kp; a.strideH = (size_t)s->Kp * s->Np;
    a.active = s->active_d;
    a.fast_divide = s->fast_divide > 0;
    a.x_in_range = s->fast_divide < 0 ? 0 : s->x_in_range;
    return a;
}
// the split path's half-steps: one launch where the owned dimension alone fills the chip, else slabs + split_apply
static int enqueue_split_h(nmf_solver *s) {
    SplitArgs a = split_args(s);
    a.nsplit = s->ns_h; a.U_out = s->H;
    { PieceScope p(s, NMF_T_H_STEP); HIPCHK(launch_split_step(a, false, s->stream)); }
    if (s->ns_h > 1) {
        PieceScope p(s, NMF_T_APPLY);
        HIPCHK(launch_split_apply(s->H, s->partials, s->vpart, s->ns_h, s->Mp, s->Np, s->Kp, a.Nv, false, s->batch, a.strideH, s->active_d, s->stream));
    }
    return NMF_OK;
}
static int enqueue_split_w(nmf_solver *s) {
    SplitArgs a = split_args(s);
    a.nsplit = s->ns_w; a.U_out = s->W;
    { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_split_step(a, true, s->stream)); }
    if (s->ns_w > 1) {
        PieceScope p(s, NMF_T_APPLY);
        HIPCHK(launch_split_apply(s->W, s->partials, s->vpart, s->ns_w, s->Mp, s->Np, s->Kp, a.Mv, true, s->batch, a.strideW, s->active_d, s->stream));
    }
    return NMF_OK;
}

// cuda/nmf.cu:118-146
static int enqueue_update_h(nmf_solver *s) {
    hipStream_t st = s->stream;
    if (s->split) return enqueue_split_h(s);
    if (s->path == NMF_PATH_FUSED) {
        if (!s->normW_fresh) { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_col_sums(s->W, s->Mp, s->Kp, s->Mp, s->normW, true, st, s->batch, (size_t)s->Mp * s->Kp)); }
        FusedArgs a = fused_args(s);
        a.nsplit = s->nsplit_h;
        if (s->nsplit_h == 1) {
            a.U_out = s->H; a.norm = s->normW; a.partial = 0;
            PieceScope p(s, NMF_T_H_STEP);
            HIPCHK(launch_fused_step(a, false, st));
        } else {
            a.partial = 1;
            { PieceScope p(s, NMF_T_H_STEP); HIPCHK(launch_fused_step(a, false, st)); }
            PieceScope p(s, NMF_T_APPLY);
            HIPCHK(launch_apply_partials(s->H, s->partials, s->nsplit_h, s->normW, s->Mp, s->Np, s->Kp, false, st, nullptr, s->batch, a.strideH, s->active_d));
        }
        return NMF_OK;
    }
    // unfused: operator-for-operator twin of cuda/nmf.cu:125-145
    const size_t mn = (size_t)s->Mp * s->Np, kn = (size_t)s->Kp * s->Np;
    {
        PieceScope p(s, NMF_T_H_STEP);
        HIPCHK(launch_gemm(GEMM_NN, s->Mp, s->Np, s->Kp, s->W, s->Mp, s->H, s->Kp, s->Z, s->Mp, st));   // nmf.cu:125
        HIPCHK(launch_set_epsilon(s->Z, mn, st));                                                       // nmf.cu:128
        HIPCHK(launch_vec_div(s->X, s->Z, s->Z, mn, st));                                               // nmf.cu:131
    }
    { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_col_sums(s->W, s->Mp, s->Kp, s->Mp, s->normW, true, st)); }   // nmf.cu:134-135
    {
        PieceScope p(s, NMF_T_H_STEP);
        HIPCHK(launch_gemm(GEMM_TN, s->Kp, s->Np, s->Mp, s->W, s->Mp, s->Z, s->Mp, s->WtZ, s->Kp, st)); // nmf.cu:138
    }
    {
        PieceScope p(s, NMF_T_APPLY);
        HIPCHK(launch_col_div(s->WtZ, s->normW, s->WtZ, s->Kp, s->Np, s->Kp, st));                      // nmf.cu:142
        HIPCHK(launch_vec_mul(s->H, s->WtZ, s->H, kn, st));                                             // nmf.cu:145
    }
    return NMF_OK;
}

// first half of cuda/nmf.cu:148-176: leaves [Z*H' ; rowsum(H)] of the local columns in psum
static int enqueue_w_partial(nmf_solver *s) {
    hipStream_t st = s->stream;
    const size_t mk = (size_t)s->Mp * s->Kp, mn = (size_t)s->Mp * s->Np;
    if (s->split) {   // raw [Z*H' ; rowsum(H)]: one forced-partial split straight into psum, or slabs + their fixed-order sum
        SplitArgs a = split_args(s);
        a.nsplit = s->ns_w; a.force_partial = 1;
        if (s->ns_w == 1) { a.partials = s->psum; a.vpart = s->psum + mk; }
        { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_split_step(a, true, st)); }
        if (s->ns_w > 1) {
            // rows of W beyond Mv hold zero padding and get no workgroup: their slab entries are whatever an H-step left in the
            // shared slab buffer, so the sum writes zeros there instead of reading them
            PieceScope p(s, NMF_T_APPLY);
            HIPCHK(launch_sum_partials(s->psum, s->partials, s->ns_w, mk, st, s->vpart, s->Kp, s->Mp, a.Mv));
        }
        return NMF_OK;
    }
    // rowsum(H) (the tail of the all-reduce operand) comes out of the W-step kernel where it can (FusedArgs::vsum_part)
    const bool vs = s->path == NMF_PATH_FUSED && fused_streams_vsum(s->Mp, s->Kp) && (s->nsplit_w == 1 || s->vsum_part);
    if (!vs) { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->psum + mk, /*clamp=*/false, st)); }
    if (s->path == NMF_PATH_FUSED) {
        FusedArgs a = fused_args(s);
        a.nsplit = s->nsplit_w; a.partial = 1;
        a.partials = (s->nsplit_w == 1) ? s->psum : s->partials;
        if (vs) a.vsum_part = (s->nsplit_w == 1) ? s->psum + mk : s->vsum_part;
        { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_fused_step(a, true, st)); }
        if (s->nsplit_w > 1) {
            PieceScope p(s, NMF_T_APPLY);
            HIPCHK(launch_sum_partials(s->psum, s->partials, s->nsplit_w, mk, st, vs ? s->vsum_part : nullptr, s->Kp));
        }
    } else {
        PieceScope p(s, NMF_T_W_STEP);
        HIPCHK(launch_gemm(GEMM_NN, s->Mp, s->Np, s->Kp, s->W, s->Mp, s->H, s->Kp, s->Z, s->Mp, st));
        HIPCHK(launch_set_epsilon(s->Z, mn, st));
        HIPCHK(launch_vec_div(s->X, s->Z, s->Z, mn, st));
        HIPCHK(launch_gemm(GEMM_NT, s->Mp, s->Kp, s->Np, s->Z, s->Mp, s->H, s->Kp, s->psum, s->Mp, st, s->partials, (size_t)s->nsplit_w * mk));
    }
    return NMF_OK;
}

static int enqueue_w_apply(nmf_solver *s) {
    const size_t mk = (size_t)s->Mp * s->Kp;
    s->normW_fresh = false;
    PieceScope p(s, NMF_T_APPLY);
    if (s->Mp <= kMaxRowsApplyColsum) {   // W *= psum / max(hsum, EPS), and the next H-step's normaliser with it (every rank: same bits)
        HIPCHK(launch_apply_w_colsum(s->W, s->psum, 1, nullptr, s->psum + mk, s->Mp, s->Kp, s->normW, s->stream));
        s->normW_fresh = true;
    } else {
        HIPCHK(launch_apply_w(s->W, s->psum, s->psum + mk, s->Mp, s->Kp, s->stream));
    }
    return NMF_OK;
}

// cuda/nmf.cu:148-176
static int enqueue_update_w(nmf_solver *s) {
    hipStream_t st = s->stream;
    s->normW_fresh = false;
    if (s->comm) {   // N-sharded: one all-reduce of (Mp*Kp + Kp) floats per iteration
        const size_t mk = (size_t)s->Mp * s->Kp;
        NMFCHK(enqueue_w_partial(s));
        { PieceScope p(s, NMF_T_ALLREDUCE); NMFCHK(nmf_comm_allreduce_f32(s->comm, s->psum, mk + (size_t)s->Kp, st)); }
        return enqueue_w_apply(s);
    }
    if (s->split) return enqueue_split_w(s);
    if (s->path == NMF_PATH_FUSED) {
        FusedArgs a = fused_args(s);
        a.nsplit = s->nsplit_w;
        if (s->split_batch > 1) {
            // (keyed on the restart count of the WHOLE call, not on this solver's share of it: a worker dealt a single restart of a
            //  batched call must run the kernels -- and the summation order of the normaliser -- its restart gets in the one-device call)
            // every pair in one launch: raw slabs (also with nsplit_w == 1), finished per pair by the apply that also leaves the next
            // H-step's normaliser.  The row sums of H come off the stream where the W-step's workgroups cover all K rows (K <= M / 16:
            // no row-sum launches), else from the two-level row-sum kernels with a pair dimension.
            a.partial = 1; a.vsum_part = s->vsum_part;
            if (!s->vsum_part) { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st, s->batch, a.strideH)); }
            { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_fused_step(a, true, st)); }
            PieceScope p(s, NMF_T_APPLY);
            HIPCHK(launch_apply_w_colsum(s->W, s->partials, s->nsplit_w, s->vsum_part ? nullptr : s->normH, s->vsum_part, s->Mp, s->Kp, s->normW, st, s->batch, a.strideW, s->active_d));
            s->normW_fresh = true;
            return NMF_OK;
        }
        if (s->nsplit_w == 1) {
            { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st)); }
            a.U_out = s->W; a.norm = s->normH; a.partial = 0;
            PieceScope p(s, NMF_T_W_STEP);
            HIPCHK(launch_fused_step(a, true, st));
        } else if (s->vsum_part) {
            // the kernel that streams H anyway also delivers its row sums (one per split); apply_partials finishes them
            a.partial = 1; a.vsum_part = s->vsum_part;
            { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_fused_step(a, true, st)); }
            PieceScope p(s, NMF_T_APPLY);
            if (s->Mp <= kMaxRowsApplyColsum) {   // the apply also leaves the next H-step's normaliser
                HIPCHK(launch_apply_w_colsum(s->W, s->partials, s->nsplit_w, nullptr, s->vsum_part, s->Mp, s->Kp, s->normW, st));
                s->normW_fresh = true;
            } else {
                HIPCHK(launch_apply_partials(s->W, s->partials, s->nsplit_w, nullptr, s->Mp, s->Np, s->Kp, true, st, s->vsum_part));
            }
        } else {
            { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st)); }
            a.partial = 1;
            { PieceScope p(s, NMF_T_W_STEP); HIPCHK(launch_fused_step(a, true, st)); }
            PieceScope p(s, NMF_T_APPLY);
            if (s->Mp <= kMaxRowsApplyColsum) {
                HIPCHK(launch_apply_w_colsum(s->W, s->partials, s->nsplit_w, s->normH, nullptr, s->Mp, s->Kp, s->normW, st));
                s->normW_fresh = true;
            } else {
                HIPCHK(launch_apply_partials(s->W, s->partials, s->nsplit_w, s->normH, s->Mp, s->Np, s->Kp, true, st));
            }
        }
        return NMF_OK;
    }
    const size_t mn = (size_t)s->Mp * s->Np, mk = (size_t)s->Mp * s->Kp;
    {
        PieceScope p(s, NMF_T_W_STEP);
        HIPCHK(launch_gemm(GEMM_NN, s->Mp, s->Np, s->Kp, s->W, s->Mp, s->H, s->Kp, s->Z, s->Mp, st));   // nmf.cu:155
        HIPCHK(launch_set_epsilon(s->Z, mn, st));                                                       // nmf.cu:158
        HIPCHK(launch_vec_div(s->X, s->Z, s->Z, mn, st));                                               // nmf.cu:161
    }
    { PieceScope p(s, NMF_T_SUMS); HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st)); }   // nmf.cu:164-165
    {
        PieceScope p(s, NMF_T_W_STEP);
        HIPCHK(launch_gemm(GEMM_NT, s->Mp, s->Kp, s->Np, s->Z, s->Mp, s->H, s->Kp, s->ZHt, s->Mp, st, s->partials, (size_t)s->nsplit_w * mk)); // nmf.cu:168
    }
    {
        PieceScope p(s, NMF_T_APPLY);
        HIPCHK(launch_row_div(s->ZHt, s->normH, s->ZHt, s->Mp, s->Kp, s->Mp, st));                      // nmf.cu:172
        HIPCHK(launch_vec_mul(s->W, s->ZHt, s->W, mk, st));                                             // nmf.cu:175
    }
    return NMF_OK;
}

extern "C" int nmf_solver_update_h(nmf_solver *s) { return s ? enqueue_update_h(s) : NMF_ERR_ARG; }
extern "C" int nmf_solver_update_w(nmf_solver *s) {
    if (!s) return NMF_ERR_ARG;
    if (s->external_reduce) { set_err("solver is in external-reduce mode: use w_partial / w_apply"); return NMF_ERR_UNSUPPORTED; }
    return enqueue_update_w(s);
}
extern "C" int nmf_solver_w_partial(nmf_solver *s) {
    if (!s) return NMF_ERR_ARG;
    if (s->batch > 1) { set_err("w_partial: batched solvers do not shard"); return NMF_ERR_UNSUPPORTED; }
    s->external_reduce = true;
    return enqueue_w_partial(s);
}
extern "C" int nmf_solver_w_apply(nmf_solver *s) { return s ? enqueue_w_apply(s) : NMF_ERR_ARG; }
extern "C" int nmf_solver_partial_buffer(nmf_solver *s, float **dev_ptr, size_t *count) {
    if (!s || !dev_ptr || !count) return NMF_ERR_ARG;
    *dev_ptr = s->psum;
    *count = (size_t)s->Mp * s->Kp + (size_t)s->Kp;
    return NMF_OK;
}

extern "C" int nmf_solver_set_partial_buffer(nmf_solver *s, float *dev_ptr, size_t count) {
    if (!s || !dev_ptr) return NMF_ERR_ARG;
    if (count < (size_t)s->Mp * s->Kp + (size_t)s->Kp) { set_err("partial buffer too small"); return NMF_ERR_ARG; }
    if (s->graph_ready) { set_err("partial buffer must be set before the first iterate()"); return NMF_ERR_UNSUPPORTED; }
    if (s->split) {   // entries of zero-padding rows are never written by the split kernel
        HIPCHK(hipMemsetAsync(dev_ptr, 0, count * sizeof(float), s->stream));
        HIPCHK(hipStreamSynchronize(s->stream));
    }
    s->psum = dev_ptr;
    return NMF_OK;
}

// cuda/nmf.cu:100-115: capture iterations, replay them.
constexpr int kGraphIters[3] = {32, 8, 1};
// One capture at a time in this process.  The capture itself is thread-local (no other thread's HIP calls are restricted by it), but
// what runs between Begin and End touches process-wide runtime state: RCCL forks its internal stream into the capture inside
// ncclAllReduce, the runtime registers and unregisters capturing streams, hipGraphInstantiate builds its own streams.  Several ranks of one
// call (nmf_multi.cpp) or several independent calls doing that side by side is the one configuration in which a HIP call on an
// unrelated stream has been seen to fail (nmf_comm.cpp: nmf_comm_wait); a capture takes 1-2 ms, so they take turns.  The lock is
// bounded: should an RCCL build ever make a captured enqueue wait for a peer rank of the same process, the peer gives up waiting
// for the lock after 5 s and captures beside it, as before.
static std::timed_mutex g_capture_mu;
static int capture_graph(nmf_solver *s, int iterations, hipGraph_t *graph, hipGraphExec_t *exec) {
    const double t0 = now_s();
    std::unique_lock<std::timed_mutex> turn(g_capture_mu, std::defer_lock);
    if (!turn.try_lock_for(std::chrono::seconds(5))) fprintf(stderr, "nmf: another thread has been capturing a hipGraph for 5 s; capturing beside it\n");
    HIPCHK(hipStreamBeginCapture(s->stream, hipStreamCaptureModeThreadLocal));
    int st = NMF_OK;
    for (int i = 0; i < iterations && st == NMF_OK; ++i) {
        st = enqueue_update_h(s);
        if (st == NMF_OK) st = enqueue_update_w(s);
    }
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(s->stream, &g);
    if (st != NMF_OK) { if (g) (void)hipGraphDestroy(g); return st; }
    if (e != hipSuccess) { set_err("hipStreamEndCapture: %s", hipGetErrorString(e)); return NMF_ERR_HIP; }
    *graph = g;
    HIPCHK(hipGraphInstantiate(exec, g, nullptr, nullptr, 0));
    s->t_setup += now_s() - t0;
    return NMF_OK;
}
static int ensure_level(nmf_solver *s, int li) {
    nmf_solver::Level &l = s->level[li];
    if (l.ready) return NMF_OK;
    if (s->comm && !s->comm_warm) {
        // RCCL sets up its transports (buffers, IPC handles, proxy connections) lazily, at the first collective of a communicator;
        // none of that may happen while a stream is capturing.  One eager all-reduce of the (scratch) partial buffer first: every
        // rank reaches this point together, the operand is overwritten by the next W-step before anything reads it.
        // It is also the first time the ranks meet: it waits with a deadline (NMF_COMM_TIMEOUT_S), and a rank that never arrives
        // gets the whole group aborted and every caller an NMF_ERR_COMM instead of a hang (update_div then falls back to one GPU
        // when sharding was its own idea).
        NMFCHK(nmf_comm_allreduce_f32(s->comm, s->psum, (size_t)s->Mp * s->Kp + (size_t)s->Kp, s->stream));
        if (nmf_comm_wait(s->comm, s->stream, nmf_comm_timeout_s(), "the first all-reduce") != NMF_OK) { set_err("the first all-reduce of the sharded run did not complete (communicator aborted)"); return NMF_ERR_COMM; }
        s->comm_warm = true;
    }
    NMFCHK(capture_graph(s, kGraphIters[li], &l.g, &l.e));
    l.ready = true;
    if (li == 2) s->graph_ready = true;
    return NMF_OK;
}

// does the W half-step as this solver runs it end in launch_apply_w_colsum (which leaves normW for the next H-step)?
static bool w_step_refreshes_normW(const nmf_solver *s) {
    if (s->Mp > kMaxRowsApplyColsum) return false;
    if (s->split) return false;   // the split kernel sums its normaliser from the factor it streams
    return s->comm ? true : (s->path == NMF_PATH_FUSED && (s->nsplit_w > 1 || s->split_batch > 1));
}

extern "C" int nmf_solver_iterate(nmf_solver *s, int iters) {
    if (!s || iters < 0) return NMF_ERR_ARG;
    if (s->external_reduce) { set_err("solver is in external-reduce mode"); return NMF_ERR_UNSUPPORTED; }
    if (iters == 0) return NMF_OK;
    if (s->use_graph && !s->timing) {
        // A captured iteration starts at the H-step.  Where the W-step's apply hands over normW, the graph holds no
        // column-sum launch and the very first H-step needs one made here; elsewhere every captured H-step brings its own.
        const bool lean = w_step_refreshes_normW(s);
        if (lean && !s->normW_fresh) {
            HIPCHK(launch_col_sums(s->W, s->Mp, s->Kp, s->Mp, s->normW, true, s->stream, s->batch, (size_t)s->Mp * s->Kp));
            s->normW_fresh = true;
        }
        if (!lean) s->normW_fresh = false;
        int st = ensure_level(s, 2);
        if (st == NMF_OK) {
            int left = iters;
            for (int li = 0; li < 3; ++li) {
                const int n = kGraphIters[li];
                if (left < n || s->level[li].failed) continue;
                if (ensure_level(s, li) != NMF_OK) {   // keep going on the shorter graphs; leave no sticky error or half-open capture behind
                    s->level[li].failed = true;
                    (void)hipGetLastError();
                    g_err[0] = 0;
                    continue;
                }
                for (; left >= n; left -= n) HIPCHK(hipGraphLaunch(s->level[li].e, s->stream));
            }
            return NMF_OK;
        }
        // capture unavailable (e.g. a collective that cannot be captured): fall back to eager launches
        fprintf(stderr, "nmf: hipGraph capture failed (%s); running eagerly\n", g_err);
        (void)hipGetLastError();
        s->use_graph = 0;
    }
    for (int i = 0; i < iters; ++i) {
        NMFCHK(enqueue_update_h(s));
        NMFCHK(enqueue_update_w(s));
    }
    return NMF_OK;
}

// capture and instantiate the graphs an iterate(iters) call would replay, without running anything: a caller that times
// iterate() (bench.py) keeps the one-off capture cost (1-2 ms) out of its timed region
extern "C" int nmf_solver_prepare(nmf_solver *s, int iters) {
    if (!s || iters < 0) return NMF_ERR_ARG;
    if (!s->use_graph || s->external_reduce) return NMF_OK;
    const bool lean = w_step_refreshes_normW(s);
    const bool fresh = s->normW_fresh;
    if (lean) s->normW_fresh = true;          // capture what iterate() captures: no column-sum launch inside the graph
    int left = iters, st = NMF_OK;
    for (int li = 0; li < 3 && st == NMF_OK; ++li) {
        const int n = kGraphIters[li];
        if (left < n || s->level[li].failed) continue;
        st = ensure_level(s, li);
        left %= n;
    }
    if (st == NMF_OK && !s->level[2].ready) st = ensure_level(s, 2);
    s->normW_fresh = fresh;
    if (st == NMF_ERR_COMM) return st;   // the communicator's first all-reduce did not complete: nothing to fall back to
    if (st != NMF_OK) { (void)hipGetLastError(); g_err[0] = 0; }   // iterate() will find out for itself and fall back
    return NMF_OK;
}

extern "C" int nmf_solver_iterate_timed(nmf_solver *s, int iters, double t[10]) {
    if (!s || iters < 0 || !t) return NMF_ERR_ARG;
    if (s->external_reduce) { set_err("solver is in external-reduce mode"); return NMF_ERR_UNSUPPORTED; }
    s->timing = true;
    int st = NMF_OK;
    for (int i = 0; i < iters && st == NMF_OK; ++i) {
        st = enqueue_update_h(s);
        if (st == NMF_OK) st = enqueue_update_w(s);
    }
    const int st2 = collect_timing(s, t);
    s->timing = false;
    return st != NMF_OK ? st : st2;
}

// KL / rel-L1 of the current state (reduce1d_div / reduce1d_diff, cuda/matrix.cu:505-640)
static int ensure_x_consts(nmf_solver *s) {
    if (s->xc_valid || !s->xc3) return NMF_OK;
    HIPCHK(launch_x_consts(s->X, (size_t)s->Mp * s->Np, s->xc_part, s->xc3, s->stream));
    s->xc_valid = true;
    return NMF_OK;
}
static int check_sums_pair(nmf_solver *s, int b, double sums[3]) {
    if (!s || !sums || b < 0 || b >= s->batch) return NMF_ERR_ARG;
    hipStream_t st = s->stream;
    {
        PieceScope p(s, NMF_T_CHECK);
        if (s->path == NMF_PATH_FUSED) {
            const float *Wb = s->W + (size_t)b * s->Mp * s->Kp, *Hb = s->H + (size_t)b * s->Kp * s->Np;
            NMFCHK(ensure_x_consts(s));
            HIPCHK(launch_check(Wb, Hb, s->X, s->Mp, s->Np, s->Kp, s->Kc, s->chk_part, st, 1, 0, 0, s->ns_chk));
            HIPCHK(launch_check_compose(s->chk_part, s->chk_groups, Wb, Hb, s->Mp, s->Np, s->Kp, s->xc3, s->sum64, s->chk_out, st));
        } else {
            const size_t mn = (size_t)s->Mp * s->Np;
            HIPCHK(launch_gemm(GEMM_NN, s->Mp, s->Np, s->Kp, s->W, s->Mp, s->H, s->Kp, s->Z, s->Mp, st));
            HIPCHK(launch_set_epsilon(s->Z, mn, st));
            HIPCHK(launch_kl_reduce(s->X, s->Z, mn, s->chk_part, st));
            HIPCHK(launch_check_final(s->chk_part, s->chk_groups, s->chk_out, st));
        }
        if (s->comm) NMFCHK(nmf_comm_allreduce_f64(s->comm, s->chk_out, 3, st));
    }
    HIPCHK(hipMemcpyAsync(s->chk_host.data(), s->chk_out, sizeof(double) * 3, hipMemcpyDeviceToHost, st));
    NMFCHK(wait_stream(s, s->block_budget_s, "a convergence check"));
    sums[0] = s->chk_host[0]; sums[1] = s->chk_host[1]; sums[2] = s->chk_host[2];
    return NMF_OK;
}
extern "C" int nmf_solver_check_sums(nmf_solver *s, double sums[3]) { return check_sums_pair(s, 0, sums); }
// every pair's three sums (KL, sum|X-WH|, sum|X|) behind one synchronisation: batch check launches back to back on the
// stream (they share the partial buffer in stream order), one copy of 3 * batch doubles
extern "C" int nmf_solver_check_all(nmf_solver *s, double *kl, double *rel_l1) {
    if (!s) return NMF_ERR_ARG;
    if (s->path != NMF_PATH_FUSED) { set_err("check_all: fused path only"); return NMF_ERR_UNSUPPORTED; }
    hipStream_t st = s->stream;
    NMFCHK(ensure_x_consts(s));
    // ONE launch of the check kernel over all pairs (blockIdx.y): a pair's check is a few workgroups (the reference's
    // 4096 x 350 x 128: six, ~130 us) and sixteen of them one after the other were 2.7 ms of a 60 ms call -- at every
    // convergence check of a run with a threshold.  The fp64 composition (three small launches) stays per pair.
    const size_t part_stride = 3 * (size_t)s->chk_groups;
    HIPCHK(launch_check(s->W, s->H, s->X, s->Mp, s->Np, s->Kp, s->Kc, s->chk_part, st, s->batch, (size_t)s->Mp * s->Kp, (size_t)s->Kp * s->Np, s->ns_chk));
    for (int b = 0; b < s->batch; ++b) {
        const float *Wb = s->W + (size_t)b * s->Mp * s->Kp, *Hb = s->H + (size_t)b * s->Kp * s->Np;
        HIPCHK(launch_check_compose(s->chk_part + (size_t)b * part_stride, s->chk_groups, Wb, Hb, s->Mp, s->Np, s->Kp, s->xc3, s->sum64, s->chk_out + 3 * (size_t)b, st));
    }
    HIPCHK(hipMemcpyAsync(s->chk_host.data(), s->chk_out, sizeof(double) * 3 * (size_t)s->batch, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    for (int b = 0; b < s->batch; ++b) {
        const double *v = s->chk_host.data() + 3 * (size_t)b;
        if (kl) kl[b] = v[0];
        if (rel_l1) rel_l1[b] = (v[2] > 0.0) ? v[1] / v[2] : 0.0;
    }
    return NMF_OK;
}
extern "C" int nmf_solver_check_pair(nmf_solver *s, int b, double *kl, double *rel_l1) {
    double v[3];
    NMFCHK(check_sums_pair(s, b, v));
    if (kl) *kl = v[0];
    if (rel_l1) *rel_l1 = (v[2] > 0.0) ? v[1] / v[2] : 0.0;
    return NMF_OK;
}
extern "C" int nmf_solver_check(nmf_solver *s, double *kl, double *rel_l1) {
    double v[3];
    NMFCHK(nmf_solver_check_sums(s, v));
    if (kl) *kl = v[0];
    if (rel_l1) *rel_l1 = (v[2] > 0.0) ? v[1] / v[2] : 0.0;
    return NMF_OK;
}

// The loop: cuda/nmf.cu:113-115 plus the README.md:40-54 convergence contract.
static int solver_run(nmf_solver *s, float thresh, int max_iter, int iter_check, int verbose, nmf_result *res, bool timed) {
    if (!s || max_iter < 0) return NMF_ERR_ARG;
    if (iter_check <= 0) iter_check = NMF_ITER_CHECK_DEFAULT;
    // verbose = 2: evaluate the checks but print nothing -- the ranks of a sharded run other than rank 0, which must issue
    // the same sequence of collectives as rank 0 (every check all-reduces its three sums)
    const bool checks = (thresh > 0.f) || verbose;
    const bool print = verbose == 1;
    s->timing = timed;
    double prev = 0.0, rl1 = 0.0;
    int nkl = 0;
    if (res) { res->n_kl = 0; res->rel_l1 = 0.0; res->path_used = s->path; }
    // Sharded: the work is handed to the device in blocks that are waited for with a deadline -- first ONE iteration under the
    // communicator's time-out (the first collectives: transports are being set up, a dead peer shows here), then blocks of
    // <= 32 under a generous multiple of what an iteration has been seen to take.
    const bool sharded = s->comm != nullptr;
    double per_iter_s = 0.0;
    s->block_budget_s = sharded ? nmf_comm_timeout_s() : 1e9;
    if (checks) {
        NMFCHK(nmf_solver_check(s, &prev, &rl1));
        if (res && nkl < NMF_MAX_KL) res->kl[nkl] = prev;
        nkl++;
        if (print) printf("iter %5d  kl-divergence %.6e  rel-L1 error %.6e\n", 0, prev, rl1);
    }
    int it = 0;
    while (it < max_iter) {
        int n = max_iter - it;
        if (checks) { const int to_check = iter_check - (it % iter_check); if (to_check < n) n = to_check; }
        if (timed && n > 256) n = 256;    // timed runs hold two hipEvents per piece: drain them block by block
        if (sharded) { const int cap = it == 0 ? 1 : 32; if (n > cap) n = cap; }
        const double tb = now_s();
        NMFCHK(nmf_solver_iterate(s, n));
        it += n;
        if (sharded) {
            NMFCHK(wait_stream(s, s->block_budget_s, it == n ? "the first iteration" : "a block of iterations"));
            const double per = (now_s() - tb) / n;
            if (per > per_iter_s) per_iter_s = per;
            s->block_budget_s = nmf_comm_timeout_s() + 100.0 * 32.0 * per_iter_s;
        }
        if (timed && s->events.size() > 2048) NMFCHK(collect_timing(s, res ? res->t : nullptr));
        if (checks && (it % iter_check) == 0) {
            double cur = 0.0;
            NMFCHK(nmf_solver_check(s, &cur, &rl1));
            if (res && nkl < NMF_MAX_KL) res->kl[nkl] = cur;
            nkl++;
            if (print) printf("iter %5d  kl-divergence %.6e  rel-L1 error %.6e  change %.3e\n", it, cur, rl1, (prev - cur) / prev);
            const bool stop = thresh > 0.f && (prev - cur) / prev < (double)thresh;   // README.md:51
            prev = cur;
            if (stop) break;
        }
    }
    NMFCHK(wait_stream(s, s->block_budget_s, "the end of the run"));
    HIPCHK(hipGetLastError());   // the reference never polls launch errors; we do
    if (res) { res->iterations = it; res->n_kl = nkl < NMF_MAX_KL ? nkl : NMF_MAX_KL; res->rel_l1 = rl1; }
    if (timed) NMFCHK(collect_timing(s, res ? res->t : nullptr));
    s->timing = false;
    return NMF_OK;
}

extern "C" int nmf_solver_run(nmf_solver *s, float thresh, int max_iter, int iter_check, int verbose, nmf_result *res) {
    if (res) memset(res->t, 0, sizeof res->t);
    return solver_run(s, thresh, max_iter, iter_check, verbose, res, false);
}

extern "C" int nmf_solver_time_piece(nmf_solver *s, int which, int reps, double *ms_per_launch) {
    if (!s || reps <= 0 || !ms_per_launch) return NMF_ERR_ARG;
    hipStream_t st = s->stream;
    hipEvent_t a, b;
    HIPCHK(hipEventCreate(&a));
    HIPCHK(hipEventCreate(&b));
    FusedArgs fa = fused_args(s);
    const size_t mk = (size_t)s->Mp * s->Kp;
    s->normW_fresh = false;   // the timed pieces overwrite W and H outside the iteration protocol
#ifdef NMF_DIAGNOSTICS
    static void *diag_scratch = nullptr;   // 128 bytes of their own for the divide census / exhaustive probes
    if (which >= 4000 && !diag_scratch) HIPCHK(hipMalloc(&diag_scratch, 128));
#endif
    // normalisers must be valid before timing an in-place fused step
    HIPCHK(launch_col_sums(s->W, s->Mp, s->Kp, s->Mp, s->normW, true, st));
    HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st));
    HIPCHK(hipEventRecord(a, st));
    for (int i = 0; i < reps; ++i) {
        switch (which) {
            case NMF_T_H_STEP:
                if (s->path != NMF_PATH_FUSED) return NMF_ERR_UNSUPPORTED;
                if (s->split) { SplitArgs sa = split_args(s); sa.nsplit = s->ns_h; sa.U_out = s->H; HIPCHK(launch_split_step(sa, false, st)); break; }
                fa.nsplit = s->nsplit_h; fa.partial = s->nsplit_h > 1; fa.U_out = s->H; fa.norm = s->normW; fa.partials = s->partials;
                HIPCHK(launch_fused_step(fa, false, st));
                break;
            case NMF_T_W_STEP:
                if (s->path != NMF_PATH_FUSED) return NMF_ERR_UNSUPPORTED;
                if (s->split) { SplitArgs sa = split_args(s); sa.nsplit = s->ns_w; sa.U_out = s->W; HIPCHK(launch_split_step(sa, true, st)); break; }
                fa.nsplit = s->nsplit_w; fa.partial = s->nsplit_w > 1; fa.U_out = s->W; fa.norm = s->normH; fa.partials = s->partials;
                HIPCHK(launch_fused_step(fa, true, st));
                break;
            case NMF_T_SUMS:
                HIPCHK(launch_col_sums(s->W, s->Mp, s->Kp, s->Mp, s->normW, true, st));
                HIPCHK(launch_row_sums(s->H, s->Kp, s->Np, s->Kp, s->rowpart, s->normH, true, st));
                break;
            case NMF_T_APPLY:
                HIPCHK(launch_sum_partials(s->psum, s->partials ? s->partials : s->psum, s->partials ? s->nsplit_w : 1, mk, st));
                break;
            case NMF_T_CHECK:
                if (s->path != NMF_PATH_FUSED) return NMF_ERR_UNSUPPORTED;
                HIPCHK(launch_check(s->W, s->H, s->X, s->Mp, s->Np, s->Kp, s->Kc, s->chk_part, st, 1, 0, 0, s->ns_chk));
                break;
            default:
#ifdef NMF_DIAGNOSTICS   // make DIAG=1; the shipped library answers NMF_ERR_ARG
                if (which >= 4001 && which <= 4004) {   // 4003: the 4-instruction variant   // 4001: slice i (of 64) of the exhaustive significand-pair comparison; 4002: rcp exponent invariance
                    if (i == 0) HIPCHK(hipMemsetAsync(diag_scratch, 0, 88, st));
                    HIPCHK(launch_divide_exhaustive((unsigned long long *)diag_scratch, which == 4002 ? -1 : (which == 4003 ? 64 + i : (which == 4004 ? 128 + i : i)), st));
                    HIPCHK(hipStreamSynchronize(st));
                    unsigned long long c[11];
                    HIPCHK(hipMemcpy(c, diag_scratch, 88, hipMemcpyDeviceToHost));
                    if (which == 4002) { fprintf(stderr, "rcp exponent invariance: %llu of %llu (significand, exponent) cases differ\n", c[0], (1ull << 23) * 123); break; }
                    fprintf(stderr, "divide exhaustive: slice %d/%d done, %llu mismatches in %llu pairs so far\n", i + 1, reps, c[0], c[1]);
                    if (i + 1 == reps) for (unsigned long long k = 0; k < c[2] && k < 8; ++k) fprintf(stderr, "    e.g. mismatch at mx=0x%06llx my=0x%06llx\n", c[3 + k] >> 32, c[3 + k] & 0xFFFFFFFFull);
                    break;
                }
                if (which == 4000) {   // IEEE vs refined-reciprocal quotient: mismatch census over ~1e9 operand pairs
                    HIPCHK(hipMemsetAsync(diag_scratch, 0, 16, st));
                    HIPCHK(launch_divide_compare((unsigned long long *)diag_scratch, 12345u + (unsigned)i, st));
                    HIPCHK(hipStreamSynchronize(st));
                    unsigned long long c2[2];
                    HIPCHK(hipMemcpy(c2, diag_scratch, 16, hipMemcpyDeviceToHost));
                    fprintf(stderr, "divide census: %llu of %llu pairs differ, max distance %llu ulp\n", c2[0], 4096ull * 256 * 1000, c2[1]);
                    break;
                }
                if (which == 3000) {   // in-kernel stamps of the v3 H-step: prints the per-chunk segment cycles
                    fa.nsplit = 1; fa.partial = 0; fa.U_out = s->H; fa.norm = s->normW; fa.partials = s->partials;
                    HIPCHK(launch_fused_stamp(fa, st));
                    HIPCHK(hipStreamSynchronize(st));
                    const int nw = ((s->Np + 127) / 128) * 4;
                    std::vector<unsigned long long> h((size_t)nw * 7);
                    HIPCHK(hipMemcpy(h.data(), s->partials, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
                    double seg[7] = {0, 0, 0, 0, 0, 0, 0};
                    for (int w = 0; w < nw; ++w) for (int q = 0; q < 7; ++q) seg[q] += (double)h[(size_t)w * 7 + q];
                    const double nchunk = (double)nw * (s->Mp / 32);
                    fprintf(stderr, "stamps (s_memtime ticks per chunk, mean over %d waves): head %.0f | product1 %.0f | divide+relayout %.0f | product2 rows 0-11 %.0f | stage_store %.0f | rows 12-15 %.0f | barrier %.0f\n",
                            nw, seg[0] / nchunk, seg[1] / nchunk, seg[2] / nchunk, seg[3] / nchunk, seg[5] / nchunk, seg[6] / nchunk, seg[4] / nchunk);
                    break;
                }
                if (which >= 2000 && which < 2003) {   // two-waves-per-SIMD partner probe
                    HIPCHK(launch_mfma_partner_probe(which - 2000, s->psum, 2000, st));
                    break;
                }
                if (which >= 1000 && which < 1100) {   // MFMA co-issue micro-probe: nv = which % 10, kind = (which - 1000) / 10
                    HIPCHK(launch_mfma_valu_probe((which - 1000) % 10, (which - 1000) / 10, s->psum, 2000, st));
                    break;
                }
                if (which >= 100 && which < 228) {   // ablation probes (timing only, clobbers H)
                    fa.nsplit = 1; fa.partial = 0; fa.U_out = s->H; fa.norm = s->normW;
                    HIPCHK(launch_fused_probe(fa, which - 100, st));
                    break;
                }
#endif
                return NMF_ERR_ARG;
        }
    }
    HIPCHK(hipEventRecord(b, st));
    HIPCHK(hipEventSynchronize(b));
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, a, b));
    (void)hipEventDestroy(a); (void)hipEventDestroy(b);
    *ms_per_launch = (double)ms / reps;
    return NMF_OK;
}

// ------------------------------------------------------------------------------ update_div
int nmf_update_div_multi(matrix W, matrix H, matrix X, const nmf_opts &o, const int *devices, int G, bool emulate, nmf_result *res);   // nmf_multi.cpp

// How many ranks, on which devices: nmf_opts.n_devices / devices / emulate_shards, NMF_DEVICES, NMF_EMULATE_SHARDS.
// `automatic` = the caller's options did not ask for several devices (n_devices = 0 and no emulate_shards): the choice came from
// the environment, and update_div_ex may fall back to one GPU if the sharded run fails.
static int plan_devices(const nmf_opts &o, const matrix &W, const matrix &H, const matrix &X, int M, int N, int K,
                        std::vector<int> &devices, bool &emulate, bool &automatic) {
    devices.clear(); emulate = false; automatic = false;
    const bool host_data = W.mat && H.mat && X.mat;
    int emu = o.emulate_shards;
    if (emu <= 1 && o.n_devices == 0 && host_data && !o.comm && !o.stream) {   // test boxes: the sharded driver without the caller asking
        const char *e = getenv("NMF_EMULATE_SHARDS");
        if (e && atoi(e) > 1 && atoi(e) <= 8 && N >= atoi(e)) { emu = atoi(e); automatic = true; }
    }
    if (emu > 1) {
        if (emu > 8 || !host_data || o.comm || o.stream || N < emu) { set_err("emulate_shards: needs host matrices, at most 8 ranks, N >= ranks, no caller stream or communicator"); return NMF_ERR_ARG; }
        int cur = 0;
        if (o.device >= 0) cur = o.device; else HIPCHK(hipGetDevice(&cur));
        devices.assign((size_t)emu, cur);
        emulate = true;
        return NMF_OK;
    }
    int want = o.n_devices;
    if (want == 1 && o.devices && host_data && !o.comm && !o.stream) {   // an explicit one-entry device list: the multi-device driver with
        devices.push_back(o.devices[0]);                                   // a single RCCL rank (what a one-GPU box can run of it)
        return NMF_OK;
    }
    if (want == 1 || want < 0) return NMF_OK;
    if (o.comm || o.stream || !host_data) {
        if (want > 1) { set_err("n_devices > 1 needs host matrices and no caller stream or communicator"); return NMF_ERR_ARG; }
        return NMF_OK;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (want == 0) {
        // Several GPUs are OPT-IN: the sharded path has never run on N > 1 real devices (no 8-GPU node was available to any round
        // of this build), and a drop-in call must not take an unproven path by itself.  n_devices = 0 therefore means ONE GPU
        // unless the environment asks: NMF_DEVICES=<n>|all forces, NMF_DEVICES=auto lets nmf_worth_sharding decide.
        automatic = true;
        // a caller that names ONE device (nmf --device 3, update_div_ex(device = 3)) gets that device
        if (o.device >= 0 && !o.devices) return NMF_OK;
        const char *e = getenv("NMF_DEVICES");
        if (!(e && e[0])) return NMF_OK;
        if (strcmp(e, "auto") == 0) want = (ndev > 1 && nmf_worth_sharding(M, N, K, ndev)) ? ndev : 1;
        else want = (strcmp(e, "all") == 0) ? ndev : atoi(e);
        if (want > ndev) want = ndev;
        if (want > N) want = N;
        if (want <= 1) return NMF_OK;
    }
    const int base = o.devices ? 0 : (o.device >= 0 ? o.device : 0);
    if (want > N || (!o.devices && base + want > ndev)) { set_err("n_devices = %d: only %d devices visible from ordinal %d (N = %d)", want, ndev, base, N); return NMF_ERR_ARG; }
    for (int g = 0; g < want; ++g) devices.push_back(o.devices ? o.devices[g] : base + g);
    return NMF_OK;
}

extern "C" int update_div_ex(matrix W, matrix H, matrix X, const nmf_opts *opts_in, nmf_result *res) {
    const double t_begin = now_s();
    nmf_opts o;
    if (opts_in) o = *opts_in; else nmf_default_opts(&o);
    if ((!W.mat && !W.mat_d) || (!H.mat && !H.mat_d) || (!X.mat && !X.mat_d)) { set_err("update_div: NULL matrix data"); return NMF_ERR_ARG; }
    const int M = W.dim[0], K = W.dim[1], N = H.dim[1];
    if (M <= 0 || K <= 0 || N <= 0) { set_err("update_div: non-positive dimension"); return NMF_ERR_ARG; }
    if (H.dim[0] != K || X.dim[0] != M || X.dim[1] != N) {
        set_err("update_div: dimensions do not agree: W %dx%d, H %dx%d, X %dx%d", W.dim[0], W.dim[1], H.dim[0], H.dim[1], X.dim[0], X.dim[1]);
        return NMF_ERR_SHAPE;
    }
    nmf_result local;
    if (!res) res = &local;
    memset(res, 0, sizeof *res);
    res->n_shards = 1; res->w_replicas_identical = 1;
    {
        std::vector<int> devices;
        bool emulate = false, automatic = false;
        NMFCHK(plan_devices(o, W, H, X, M, N, K, devices, emulate, automatic));
        if (!devices.empty()) {
            const int st = nmf_update_div_multi(W, H, X, o, devices.data(), (int)devices.size(), emulate, res);
            if (!(st == NMF_ERR_COMM && automatic)) return st;
            // Sharding was the library's own idea and it failed (no communicator, a rank that never reached the first collective,
            // a collective that broke mid-run).  nmf_update_div_multi writes W.mat / H.mat only after EVERY rank has succeeded,
            // so the caller's factors are still the initial ones: run the whole problem on one GPU, in this process.
            fprintf(stderr, "nmf: the sharded run over %d %s failed (%s); running on one GPU\n", (int)devices.size(), emulate ? "emulated shards" : "devices", nmf_last_error());
            memset(res, 0, sizeof *res);
            res->n_shards = 1; res->w_replicas_identical = 1;
        }
    }
    // A one-shot call pays for capture and instantiation (1-2 ms) out of its own run time: worth it only for long runs -- every
    // launch of this path outlasts its own host-side enqueue, so replay does not make a short run faster (cfg2, 200 iterations:
    // 12.7 ms captured, 12.0 ms eager; the paper's shape 9.9 / 7.8 ms).  The resident solver (nmf_solver_*) keeps its graphs.
    if (o.use_graph == NMF_GRAPH_AUTO) o.use_graph = (8.0 * M * N * K * (double)o.max_iter < 2e12 && !o.comm) ? -1 : 1;
    nmf_solver *s = nullptr;
    NMFCHK(nmf_solver_create(&s, M, N, K, &o));
    int st = NMF_OK;
    double t0 = now_s();
    // host data wins when present; otherwise the caller's device buffers are the source
    if (st == NMF_OK) st = W.mat ? nmf_solver_upload(s, W.mat, nullptr, nullptr) : nmf_solver_upload_device(s, W.mat_d, nullptr, nullptr);
    if (st == NMF_OK) st = H.mat ? nmf_solver_upload(s, nullptr, H.mat, nullptr) : nmf_solver_upload_device(s, nullptr, H.mat_d, nullptr);
    if (st == NMF_OK) st = X.mat ? nmf_solver_upload(s, nullptr, nullptr, X.mat) : nmf_solver_upload_device(s, nullptr, nullptr, X.mat_d);
    if (st == NMF_OK) st = nmf_solver_sync(s);
    res->t[NMF_T_H2D] = now_s() - t0;
    // use_graph == 0 -> eager launches, each piece bracketed by hipEvents (fills t[2..7])
    const bool timed = (o.use_graph == 0);
    if (st == NMF_OK) st = solver_run(s, o.converge_thresh, o.max_iter, o.iter_check, o.verbose, res, timed);
    t0 = now_s();
    if (st == NMF_OK) {
        if (W.mat) st = nmf_solver_download(s, W.mat, nullptr);
        if (st == NMF_OK && H.mat) st = nmf_solver_download(s, nullptr, H.mat);
        // keep caller-provided device mirrors coherent (unpadded layout)
        if (st == NMF_OK && W.mat_d) {
            if (launch_unpad_copy(W.mat_d, M, K, s->W, s->Mp, s->stream) != hipSuccess) st = NMF_ERR_HIP;
        }
        if (st == NMF_OK && H.mat_d) {
            if (launch_unpad_copy(H.mat_d, K, N, s->H, s->Kp, s->stream) != hipSuccess) st = NMF_ERR_HIP;
        }
        if (st == NMF_OK) st = nmf_solver_sync(s);
    }
    res->t[NMF_T_D2H] = now_s() - t0;
    res->t[NMF_T_SETUP] = s->t_setup;
    nmf_solver_destroy(s);
    res->t[NMF_T_TOTAL] = now_s() - t_begin;
    return st;
}

// Multi-restart (paper section 3.2): X resident once, every (W,H) pair through the same loop, best final KL wins.
// Several solvers iterate side by side, each on its own stream, with solver_run's convergence logic applied per lane:
// the enqueue of a block of iterations is asynchronous, so the lanes' kernels overlap on the device.
static int run_lanes(std::vector<nmf_solver *> &lane, int n, float thresh, int max_iter, int iter_check, int verbose, int first_index, const int *gidx = nullptr) {
    if (iter_check <= 0) iter_check = NMF_ITER_CHECK_DEFAULT;
    const bool checks = (thresh > 0.f) || verbose;
    std::vector<double> prev((size_t)n, 0.0);
    std::vector<char> done((size_t)n, 0);
    double rl1 = 0.0;
    if (checks)
        for (int l = 0; l < n; ++l) {
            NMFCHK(nmf_solver_check(lane[l], &prev[l], &rl1));
            if (verbose) printf("restart %d iter %5d  kl-divergence %.6e  rel-L1 error %.6e\n", gidx ? gidx[first_index + l] : first_index + l, 0, prev[l], rl1);
        }
    for (int it = 0; it < max_iter;) {
        int nstep = max_iter - it;
        if (checks) { const int to_check = iter_check - (it % iter_check); if (to_check < nstep) nstep = to_check; }
        bool any = false;
        for (int l = 0; l < n; ++l) if (!done[l]) { NMFCHK(nmf_solver_iterate(lane[l], nstep)); any = true; }
        if (!any) break;
        it += nstep;
        if (checks && (it % iter_check) == 0)
            for (int l = 0; l < n; ++l) {
                if (done[l]) continue;
                double cur = 0.0;
                NMFCHK(nmf_solver_check(lane[l], &cur, &rl1));
                if (verbose) printf("restart %d iter %5d  kl-divergence %.6e  rel-L1 error %.6e  change %.3e\n", gidx ? gidx[first_index + l] : first_index + l, it, cur, rl1, (prev[l] - cur) / prev[l]);
                if (thresh > 0.f && (prev[l] - cur) / prev[l] < (double)thresh) done[l] = 1;   // README.md:51
                prev[l] = cur;
            }
    }
    for (int l = 0; l < n; ++l) { HIPCHK(hipStreamSynchronize(lane[l]->stream)); }
    HIPCHK(hipGetLastError());
    return NMF_OK;
}

// Measured (tools/restart_bench.py): two lanes overlap one lane's launch gaps and small grids with the other's work
// (1.3-1.5x on 1024 x 4096 x 64 and 512 x 3445 x 30); more lanes only queue up behind the command processor.  A launch
// of more than 512 workgroups fills the chip by itself.
static int auto_lanes(const nmf_solver *s, int n_restarts) {
    const int per_group = (s->path == NMF_PATH_FUSED) ? fused_cols_per_group(s->Kp) : 128;
    const long groups = (long)((s->Np + per_group - 1) / per_group) * (s->nsplit_h > 0 ? s->nsplit_h : 1);
    const int lanes = groups <= 512 ? 2 : 1;
    return lanes > n_restarts ? n_restarts : lanes;
}

// The restarts as ONE batched solver: the restart index is a grid dimension of the split kernel, so every launch carries
// all B pairs (B x the workgroups; X tiles shared through L2), each pair iterating exactly as a sequential update_div on
// it would -- same kernels, same split counts, hence the same bits -- and freezing at its own convergence check.
constexpr int kMaxRestartBatch = 64;
static int restarts_batched(const matrix *W, const matrix *H, int n_restarts, matrix X, const nmf_opts &o_in, int M, int N, int K, int *best, double *kl, int split_batch, const int *gidx = nullptr) {
    int B = n_restarts < kMaxRestartBatch ? n_restarts : kMaxRestartBatch;
    nmf_opts o = o_in;
    // as in update_div_ex: a run this short does not earn back the capture and instantiation of its graphs (every launch of a
    // batch outlasts its own enqueue by far): 16 restarts x 200 iterations, whole call, eager / captured: cfg2 69.5-70.3 /
    // 70.4-71.0 ms, gold 56.9-59.0 / 58.9-59.1, paper 21.1 / 22.9-23.1 (tools/restart_graph_ab.py, round 3)
    if (o.use_graph == NMF_GRAPH_AUTO) o.use_graph = (8.0 * M * N * K * (double)o.max_iter < 2e12) ? -1 : 1;
    const bool trace = getenv("NMF_RESTART_TRACE") != nullptr;   // wall time of every phase of the call, to stderr
    double tp = now_s();
    auto phase = [&](const char *what) { if (trace) { const double t = now_s(); fprintf(stderr, "nmf restarts: %-28s %8.3f ms\n", what, (t - tp) * 1e3); tp = t; } };
    // A batched solver holds B copies of W, H and of the slabs behind ONE X (65536 x 512 with 64 pairs: > 8 GB of slabs per cut).  Where
    // the device cannot hold B pairs the call makes more passes with fewer: the cuts -- hence every restart's bits -- follow split_batch,
    // the restart count of the whole call, not the number of pairs one solver carries (as with the 64-pair limit and with worker shares).
    nmf_solver *s = nullptr;
    for (;;) {
        const int cst = create_batched(&s, M, N, K, B, &o, split_batch);
        if (cst == NMF_OK) break;
        if (cst != NMF_ERR_NOMEM || B == 1) return cst;
        (void)hipGetLastError();
        B = (B + 1) / 2;
        if (trace) fprintf(stderr, "nmf restarts: not enough device memory for the batch; retrying with %d pairs per pass\n", B);
    }
    phase("solver (allocation)");
    int st = X.mat ? nmf_solver_upload(s, nullptr, nullptr, X.mat) : nmf_solver_upload_device(s, nullptr, nullptr, X.mat_d);
    phase("X upload");
    const int iter_check = o.iter_check > 0 ? o.iter_check : NMF_ITER_CHECK_DEFAULT;
    const bool checks = (o.converge_thresh > 0.f) || o.verbose;
    std::vector<double> prev((size_t)B), cur((size_t)B), rl1((size_t)B);
    std::vector<int> act((size_t)B);
    int best_i = -1;
    double best_kl = 0.0;
    for (int base = 0; st == NMF_OK && base < n_restarts; base += B) {
        const int n = (n_restarts - base < B) ? (n_restarts - base) : B;
        if (st == NMF_OK) st = upload_pairs(s, W + base, H + base, n);
        for (int b = 0; b < B; ++b) act[(size_t)b] = b < n;
        if (st == NMF_OK && (n < B || base > 0)) st = nmf_solver_set_active(s, act.data());   // a full first batch: every flag is still 1
        phase("pair uploads");
        if (st == NMF_OK && checks) {
            st = nmf_solver_check_all(s, prev.data(), rl1.data());
            if (o.verbose) for (int b = 0; b < n; ++b) printf("restart %d iter %5d  kl-divergence %.6e  rel-L1 error %.6e\n", gidx ? gidx[base + b] : base + b, 0, prev[(size_t)b], rl1[(size_t)b]);
        }
        for (int it = 0; st == NMF_OK && it < o.max_iter;) {
            int nstep = o.max_iter - it;
            if (checks) { const int to_check = iter_check - (it % iter_check); if (to_check < nstep) nstep = to_check; }
            bool any = false;
            for (int b = 0; b < n; ++b) any = any || act[(size_t)b];
            if (!any) break;
            st = nmf_solver_iterate(s, nstep);
            it += nstep;
            if (st == NMF_OK && checks && (it % iter_check) == 0) {
                st = nmf_solver_check_all(s, cur.data(), rl1.data());
                bool changed = false;
                for (int b = 0; st == NMF_OK && b < n; ++b) {
                    if (!act[(size_t)b]) continue;
                    const double p = prev[(size_t)b], c = cur[(size_t)b];
                    if (o.verbose) printf("restart %d iter %5d  kl-divergence %.6e  rel-L1 error %.6e  change %.3e\n", gidx ? gidx[base + b] : base + b, it, c, rl1[(size_t)b], (p - c) / p);
                    if (o.converge_thresh > 0.f && (p - c) / p < (double)o.converge_thresh) { act[(size_t)b] = 0; changed = true; }   // README.md:51
                    prev[(size_t)b] = c;
                }
                if (st == NMF_OK && changed) st = nmf_solver_set_active(s, act.data());
            }
        }
        if (trace && st == NMF_OK) { st = nmf_solver_sync(s); phase("iterations"); }
        if (st == NMF_OK) st = nmf_solver_check_all(s, cur.data(), nullptr);
        phase("final KL of every pair");
        if (st == NMF_OK) st = download_pairs(s, W + base, H + base, n);
        phase("pair downloads");
        for (int b = 0; st == NMF_OK && b < n; ++b) {
            if (kl) kl[base + b] = cur[(size_t)b];
            if (best_i < 0 || cur[(size_t)b] < best_kl) { best_i = base + b; best_kl = cur[(size_t)b]; }
        }
    }
    if (st == NMF_OK && hipGetLastError() != hipSuccess) { set_err("update_div_restarts: a kernel launch failed"); st = NMF_ERR_HIP; }
    nmf_solver_destroy(s);
    phase("solver destroyed");
    if (st == NMF_OK && best) *best = best_i;
    return st;
}

// Do the restarts of this shape run as ONE batched solver (the restart index a grid dimension)?  Shapes the split kernel takes:
// always.  Shapes of the 64-column kernel: where a lone pair's launch leaves CUs idle or needs partial slabs to fill them --
// fewer than 512 column groups in the H-step (N < 32768) -- e.g. 4096 x 4096 x 256, 2048 x 8192 x 512; above that one launch
// fills the chip by itself and the restarts run one after the other (auto_lanes).
static bool restarts_take_batch(int M, int N, int K, const nmf_opts &o, int cus) {
    if (want_split(M, N, K, o, cus)) return true;
    if (o.path == NMF_PATH_UNFUSED || o.split_kernel > 0) return false;
    const int kp = fused_pad_k(K), Mp = pad32(M), Np = pad32(N);
    if (!kp || !fused_takes_batch(kp) || Mp > kMaxRowsApplyColsum || (size_t)kp * (size_t)Mp >= ((size_t)1 << 31)) return false;
    return (Np + 63) / 64 < 2 * cus;
}

// all restarts of `W`, `H` on ONE device (o.device, or the current one); split_batch: restart count of the whole call
static int restarts_one_device(const matrix *W, const matrix *H, int n_restarts, matrix X, const nmf_opts &o, int M, int N, int K, int *best, double *kl, int split_batch, const int *gidx = nullptr) {
    if (o.device >= 0) HIPCHK(hipSetDevice(o.device));
    // restart_lanes > 0 asks for the stream-lane mechanism explicitly (the shapes no batched kernel takes use it anyway)
    const bool batched = o.restart_lanes <= 0 && !o.comm && restarts_take_batch(M, N, K, o, device_cus());
    if (n_restarts > 1 && batched) return restarts_batched(W, H, n_restarts, X, o, M, N, K, best, kl, split_batch, gidx);
    if (n_restarts == 1 && split_batch > 1 && batched)   // this device's share of a batched call
        return restarts_batched(W, H, 1, X, o, M, N, K, best, kl, split_batch, gidx);
    std::vector<nmf_solver *> lane;
    nmf_solver *s0 = nullptr;
    NMFCHK(nmf_solver_create(&s0, M, N, K, &o));
    lane.push_back(s0);
    int st = X.mat ? nmf_solver_upload(s0, nullptr, nullptr, X.mat) : nmf_solver_upload_device(s0, nullptr, nullptr, X.mat_d);
    int lanes = o.restart_lanes > 0 ? o.restart_lanes : auto_lanes(s0, n_restarts);
    if (lanes > n_restarts) lanes = n_restarts;
    if (o.stream || o.comm) lanes = 1;            // a caller-owned stream or communicator cannot be shared between lanes
    nmf_opts ol = o;
    ol.stream = nullptr;
    for (int l = 1; st == NMF_OK && l < lanes; ++l) {   // further lanes share s0's X
        nmf_solver *s = new nmf_solver();
        st = solver_init(s, M, N, K, ol, s0);
        lane.push_back(s);
    }
    int best_i = -1;
    double best_kl = 0.0;
    for (int base = 0; st == NMF_OK && base < n_restarts; base += lanes) {
        const int n = (n_restarts - base < lanes) ? (n_restarts - base) : lanes;
        for (int l = 0; st == NMF_OK && l < n; ++l) st = nmf_solver_upload(lane[l], W[base + l].mat, H[base + l].mat, nullptr);
        if (st == NMF_OK) st = run_lanes(lane, n, o.converge_thresh, o.max_iter, o.iter_check, o.verbose, base, gidx);
        for (int l = 0; st == NMF_OK && l < n; ++l) {
            double v = 0.0;
            st = nmf_solver_check(lane[l], &v, nullptr);
            if (st == NMF_OK) st = nmf_solver_download(lane[l], W[base + l].mat, H[base + l].mat);
            if (st == NMF_OK) {
                if (kl) kl[base + l] = v;
                if (best_i < 0 || v < best_kl) { best_i = base + l; best_kl = v; }
            }
        }
    }
    for (size_t l = lane.size(); l-- > 0;) nmf_solver_destroy(lane[l]);   // s0 (the owner of X) last
    if (st == NMF_OK && best) *best = best_i;
    return st;
}

// Which devices share the restarts ("replicas only": SURVEY 8e / 8f4 -- below the size where sharding one problem pays, a node's
// GPUs are used by giving each its own restarts; no communicator, no collective, nothing to wait for but the threads).
// nmf_opts.n_devices = n > 1 (with an optional `devices` list, which may name a device more than once: each entry is a worker
// with its own solver and stream) forces; 0 = one device unless NMF_DEVICES=<n>|all|auto asks for more (and the caller did not
// pin one with device >= 0); 1 = the one device.
static void plan_restart_devices(const nmf_opts &o, int n_restarts, bool host_x, double flop, std::vector<int> &devices) {
    devices.clear();
    if (o.comm || o.stream || !host_x || n_restarts < 2) return;
    int want = o.n_devices;
    if (want == 1 || want < 0) return;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    if (want == 0) {   // opt-in, as for update_div: one device unless NMF_DEVICES=<n>|all|auto asks for more
        if (o.device >= 0 && !o.devices) return;
        const char *e = getenv("NMF_DEVICES");
        if (!(e && e[0])) return;
        // auto: the library spreads only work worth it -- bringing up a device this process has not used yet costs a few hundred
        // milliseconds, so below ~0.5 s of single-GPU work (5e13 flop) the restarts stay on one device
        if (strcmp(e, "auto") == 0) { if (flop < 5e13) return; want = ndev; }
        else want = (strcmp(e, "all") == 0) ? ndev : atoi(e);
        if (want > ndev) want = ndev;
    }
    if (want > n_restarts) want = n_restarts;
    if (want <= 1) return;
    const int base = o.devices ? 0 : (o.device >= 0 ? o.device : 0);
    for (int g = 0; g < want; ++g) {
        const int d = o.devices ? o.devices[g] : base + g;
        if (d < 0 || d >= ndev) { devices.clear(); return; }
        devices.push_back(d);
    }
}

extern "C" int update_div_restarts(const matrix *W, const matrix *H, int n_restarts, matrix X, const nmf_opts *opts_in, int *best, double *kl) {
    if (!W || !H || n_restarts < 1 || (!X.mat && !X.mat_d)) { set_err("update_div_restarts: bad arguments"); return NMF_ERR_ARG; }
    nmf_opts o;
    if (opts_in) o = *opts_in; else nmf_default_opts(&o);
    const int M = X.dim[0], N = X.dim[1], K = W[0].dim[1];
    for (int i = 0; i < n_restarts; ++i) {
        if (!W[i].mat || !H[i].mat) { set_err("update_div_restarts: pair %d has no host data", i); return NMF_ERR_ARG; }
        if (W[i].dim[0] != M || W[i].dim[1] != K || H[i].dim[0] != K || H[i].dim[1] != N) {
            set_err("update_div_restarts: pair %d: dimensions do not agree", i);
            return NMF_ERR_SHAPE;
        }
    }
    if (o.n_devices > 1 && (o.comm || o.stream || !X.mat)) { set_err("update_div_restarts: n_devices > 1 needs a host X and no caller stream or communicator"); return NMF_ERR_ARG; }
    const int split_batch = n_restarts < kMaxRestartBatch ? n_restarts : kMaxRestartBatch;
    std::vector<int> devices;
    plan_restart_devices(o, n_restarts, X.mat != nullptr, 8.0 * M * N * K * (double)o.max_iter * n_restarts, devices);
    if (o.n_devices > 1 && devices.empty() && n_restarts > 1) { set_err("update_div_restarts: n_devices = %d: not that many devices visible", o.n_devices); return NMF_ERR_ARG; }
    if (devices.empty()) return restarts_one_device(W, H, n_restarts, X, o, M, N, K, best, kl, split_batch);

    // Restart i goes to worker i % G: one host thread, one (batched) solver, one upload of X per worker.  Every restart runs the
    // same kernels with the same split counts as in a one-device call (split_batch), so its result has the same bits.
    const int G = (int)devices.size();
    struct Worker { std::vector<matrix> W, H; std::vector<int> idx; std::vector<double> kl; int status = NMF_OK; char err[512] = ""; };
    std::vector<Worker> wk((size_t)G);
    for (int i = 0; i < n_restarts; ++i) { Worker &w = wk[(size_t)(i % G)]; w.W.push_back(W[i]); w.H.push_back(H[i]); w.idx.push_back(i); }
    std::vector<std::thread> th;
    for (int g = 0; g < G; ++g)
        th.emplace_back([&, g]() {
            Worker &w = wk[(size_t)g];
            nmf_opts og = o;
            og.device = devices[(size_t)g]; og.n_devices = 1; og.devices = nullptr; og.emulate_shards = 0;
            w.kl.assign(w.idx.size(), 0.0);
            int b = -1;
            w.status = restarts_one_device(w.W.data(), w.H.data(), (int)w.idx.size(), X, og, M, N, K, &b, w.kl.data(), split_batch, w.idx.data());
            if (w.status != NMF_OK) snprintf(w.err, sizeof w.err, "device %d: %s", devices[(size_t)g], nmf_last_error());
        });
    for (auto &t : th) t.join();
    int best_i = -1;
    double best_kl = 0.0;
    for (int g = 0; g < G; ++g) if (wk[(size_t)g].status != NMF_OK) { set_err("%s", wk[(size_t)g].err); return wk[(size_t)g].status; }
    for (int i = 0; i < n_restarts; ++i) {   // in restart order: the lowest index wins a tie, as on one device
        const double v = wk[(size_t)(i % G)].kl[(size_t)(i / G)];
        if (kl) kl[i] = v;
        if (best_i < 0 || v < best_kl) { best_i = i; best_kl = v; }
    }
    if (best) *best = best_i;
    return NMF_OK;
}

// README.md:40-46.  void + exit on error like the reference (error-check.hpp:12-17).
extern "C" void update_div(matrix W, matrix H, matrix X, float CONVERGE_THRESH, int max_iter, double t[10], int verbose) {
    nmf_opts o;
    nmf_default_opts(&o);
    o.converge_thresh = CONVERGE_THRESH;
    o.max_iter = max_iter;
    o.verbose = verbose;
    o.use_graph = (t == nullptr) ? NMF_GRAPH_AUTO : 0;   // timers requested: eager launches with per-piece hipEvents
    nmf_result res;
    const int st = update_div_ex(W, H, X, &o, &res);
    if (st != NMF_OK) {
        fprintf(stderr, "update_div: %s (%s)\n", nmf_status_string(st), nmf_last_error());
        exit(st);
    }
    if (t) for (int i = 0; i < 10; ++i) t[i] = res.t[i];
    if (verbose) printf("update_div: %d iterations, %.3f s total\n", res.iterations, res.t[NMF_T_TOTAL]);
}

// ------------------------------------------------------------------------------ operators
static int need_dev(const matrix &a, const char *who) {
    if (!a.mat_d) { set_err("%s: matrix has no device buffer (mat_d == NULL)", who); return NMF_ERR_ARG; }
    if (a.dim[0] <= 0 || a.dim[1] <= 0) { set_err("%s: non-positive dimension", who); return NMF_ERR_ARG; }
    return NMF_OK;
}
static int shape_err(const char *who) {
    set_err("%s: dimensions do not agree", who);
    return NMF_ERR_SHAPE;
}

extern "C" int nmf_matrix_multiply(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "matrix_multiply")); NMFCHK(need_dev(b, "matrix_multiply")); NMFCHK(need_dev(c, "matrix_multiply"));
    if (a.dim[1] != b.dim[0] || c.dim[0] != a.dim[0] || c.dim[1] != b.dim[1]) return shape_err("matrix_multiply");
    HIPCHK(launch_gemm(GEMM_NN, c.dim[0], c.dim[1], a.dim[1], a.mat_d, a.dim[0], b.mat_d, b.dim[0], c.mat_d, c.dim[0], (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_matrix_multiply_AtB(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "matrix_multiply_AtB")); NMFCHK(need_dev(b, "matrix_multiply_AtB")); NMFCHK(need_dev(c, "matrix_multiply_AtB"));
    if (a.dim[0] != b.dim[0] || c.dim[0] != a.dim[1] || c.dim[1] != b.dim[1]) return shape_err("matrix_multiply_AtB");
    HIPCHK(launch_gemm(GEMM_TN, c.dim[0], c.dim[1], a.dim[0], a.mat_d, a.dim[0], b.mat_d, b.dim[0], c.mat_d, c.dim[0], (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_matrix_multiply_ABt(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "matrix_multiply_ABt")); NMFCHK(need_dev(b, "matrix_multiply_ABt")); NMFCHK(need_dev(c, "matrix_multiply_ABt"));
    if (a.dim[1] != b.dim[1] || c.dim[0] != a.dim[0] || c.dim[1] != b.dim[0]) return shape_err("matrix_multiply_ABt");
    // small output + long reduction: split K through a temporary slab workspace (this entry point synchronises then)
    const size_t per = (size_t)c.dim[0] * c.dim[1];
    const int tiles = ((c.dim[0] + 127) / 128) * ((c.dim[1] + 127) / 128);
    if (tiles < 128 && a.dim[1] >= 4096) {
        float *ws = nullptr;
        const size_t nws = 16 * per;
        HIPCHK(hipMalloc((void **)&ws, nws * sizeof(float)));
        hipError_t e = launch_gemm(GEMM_NT, c.dim[0], c.dim[1], a.dim[1], a.mat_d, a.dim[0], b.mat_d, b.dim[0], c.mat_d, c.dim[0], (hipStream_t)stream, ws, nws);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        (void)hipFree(ws);
        HIPCHK(e);
        return NMF_OK;
    }
    HIPCHK(launch_gemm(GEMM_NT, c.dim[0], c.dim[1], a.dim[1], a.mat_d, a.dim[0], b.mat_d, b.dim[0], c.mat_d, c.dim[0], (hipStream_t)stream));
    return NMF_OK;
}
static bool same_shape(const matrix &a, const matrix &b) { return a.dim[0] == b.dim[0] && a.dim[1] == b.dim[1]; }

extern "C" int nmf_element_multiply(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "element_multiply")); NMFCHK(need_dev(b, "element_multiply")); NMFCHK(need_dev(c, "element_multiply"));
    if (!same_shape(a, b) || !same_shape(a, c)) return shape_err("element_multiply");
    HIPCHK(launch_vec_mul(a.mat_d, b.mat_d, c.mat_d, (size_t)a.dim[0] * a.dim[1], (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_element_divide(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "element_divide")); NMFCHK(need_dev(b, "element_divide")); NMFCHK(need_dev(c, "element_divide"));
    if (!same_shape(a, b) || !same_shape(a, c)) return shape_err("element_divide");
    HIPCHK(launch_vec_div(a.mat_d, b.mat_d, c.mat_d, (size_t)a.dim[0] * a.dim[1], (hipStream_t)stream));
    return NMF_OK;
}
// every row of a divided element-wise by the vector b (length = a's column count): cuda/matrix.cu:203-224
extern "C" int nmf_row_divide(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "row_divide")); NMFCHK(need_dev(b, "row_divide")); NMFCHK(need_dev(c, "row_divide"));
    if (a.dim[1] != b.dim[0] || b.dim[1] != 1 || !same_shape(a, c)) return shape_err("row_divide");
    HIPCHK(launch_row_div(a.mat_d, b.mat_d, c.mat_d, a.dim[0], a.dim[1], a.dim[0], (hipStream_t)stream));
    return NMF_OK;
}
// every column of a divided element-wise by the vector b (length = a's row count): cuda/matrix.cu:226-250
extern "C" int nmf_col_divide(matrix a, matrix b, matrix c, void *stream) {
    NMFCHK(need_dev(a, "col_divide")); NMFCHK(need_dev(b, "col_divide")); NMFCHK(need_dev(c, "col_divide"));
    if (a.dim[0] != b.dim[1] || b.dim[0] != 1 || !same_shape(a, c)) return shape_err("col_divide");
    HIPCHK(launch_col_div(a.mat_d, b.mat_d, c.mat_d, a.dim[0], a.dim[1], a.dim[0], (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_set_epsilon(matrix a, void *stream) {
    NMFCHK(need_dev(a, "set_epsilon"));
    HIPCHK(launch_set_epsilon(a.mat_d, (size_t)a.dim[0] * a.dim[1], (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_sum_cols(matrix a, matrix out, void *stream) {
    NMFCHK(need_dev(a, "sum_cols")); NMFCHK(need_dev(out, "sum_cols"));
    if (out.dim[0] != 1 || out.dim[1] != a.dim[1]) return shape_err("sum_cols");   // cuda/matrix.cu:272-275
    HIPCHK(launch_col_sums(a.mat_d, a.dim[0], a.dim[1], a.dim[0], out.mat_d, false, (hipStream_t)stream));
    return NMF_OK;
}
extern "C" int nmf_sum_rows(matrix a, matrix out, void *stream) {
    NMFCHK(need_dev(a, "sum_rows")); NMFCHK(need_dev(out, "sum_rows"));
    if (out.dim[1] != 1 || out.dim[0] != a.dim[0]) return shape_err("sum_rows");   // cuda/matrix.cu:391-394
    float *part = nullptr;
    HIPCHK(hipMalloc((void **)&part, sizeof(float) * (size_t)row_sum_blocks(a.dim[1]) * a.dim[0]));
    hipError_t e = launch_row_sums(a.mat_d, a.dim[0], a.dim[1], a.dim[0], part, out.mat_d, false, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(part);
    HIPCHK(e);
    return NMF_OK;
}

static int reduce3(matrix x, matrix y, double out3[3], void *stream, const char *who) {
    NMFCHK(need_dev(x, who)); NMFCHK(need_dev(y, who));
    if (!same_shape(x, y)) return shape_err(who);
    const size_t n = (size_t)x.dim[0] * x.dim[1];
    const int groups = reduce_num_groups(n);
    double *part = nullptr, *out = nullptr;
    HIPCHK(hipMalloc((void **)&part, sizeof(double) * 3 * (size_t)groups));
    HIPCHK(hipMalloc((void **)&out, sizeof(double) * 3));
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = launch_kl_reduce(x.mat_d, y.mat_d, n, part, st);
    if (e == hipSuccess) e = launch_check_final(part, groups, out, st);
    if (e == hipSuccess) e = hipMemcpyAsync(out3, out, sizeof(double) * 3, hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(part); (void)hipFree(out);
    HIPCHK(e);
    return NMF_OK;
}
extern "C" int nmf_kl_divergence(matrix x, matrix y, double *kl, void *stream) {
    if (!kl) return NMF_ERR_ARG;
    double o[3];
    NMFCHK(reduce3(x, y, o, stream, "kl_divergence"));
    *kl = o[0];
    return NMF_OK;
}
extern "C" int nmf_diff_norm(matrix x, matrix y, double *sum_abs_diff, double *sum_abs_x, void *stream) {
    double o[3];
    NMFCHK(reduce3(x, y, o, stream, "diff_norm"));
    if (sum_abs_diff) *sum_abs_diff = o[1];
    if (sum_abs_x) *sum_abs_x = o[2];
    return NMF_OK;
}

// ------------------------------------------------------------------------------ matrix helpers
extern "C" int nmf_create_matrix(matrix *A, int rows, int cols, float value) {
    if (!A || rows <= 0 || cols <= 0) return NMF_ERR_ARG;
    const size_t n = (size_t)rows * cols;
    A->mat = (float *)malloc(n * sizeof(float));
    if (!A->mat) return NMF_ERR_NOMEM;
    for (size_t i = 0; i < n; ++i) A->mat[i] = value;
    A->mat_d = nullptr;
    A->dim[0] = rows; A->dim[1] = cols;
    return NMF_OK;
}
// Matrix(rows, cols), cuda/matrix.cu:42-51: a device buffer of that shape and nothing else (no host copy; contents undefined)
extern "C" int nmf_matrix_alloc_device(matrix *A, int rows, int cols) {
    if (!A || rows <= 0 || cols <= 0) return NMF_ERR_ARG;
    A->mat = nullptr; A->mat_d = nullptr;
    A->dim[0] = rows; A->dim[1] = cols;
    HIPCHK(hipMalloc((void **)&A->mat_d, (size_t)rows * cols * sizeof(float)));
    return NMF_OK;
}
extern "C" int nmf_matrix_free_device(matrix *A) {
    if (!A) return NMF_ERR_ARG;
    if (A->mat_d) { HIPCHK(hipFree(A->mat_d)); A->mat_d = nullptr; }
    return NMF_OK;
}
extern "C" void nmf_destroy_matrix(matrix *A) {
    if (!A) return;
    if (A->mat) free(A->mat);
    A->mat = nullptr;
    if (A->mat_d) (void)hipFree(A->mat_d);
    A->mat_d = nullptr;
    A->dim[0] = A->dim[1] = 0;
}
extern "C" int nmf_matrix_to_device(matrix *A) {
    if (!A || !A->mat || A->dim[0] <= 0 || A->dim[1] <= 0) return NMF_ERR_ARG;
    const size_t bytes = (size_t)A->dim[0] * A->dim[1] * sizeof(float);   // 64-bit sizes (cf. uint32 overflow, cuda/matrix.cu:61)
    if (!A->mat_d) HIPCHK(hipMalloc((void **)&A->mat_d, bytes));
    HIPCHK(hipMemcpy(A->mat_d, A->mat, bytes, hipMemcpyHostToDevice));
    return NMF_OK;
}
extern "C" int nmf_matrix_from_device(matrix *A) {
    if (!A || !A->mat || !A->mat_d) return NMF_ERR_ARG;
    const size_t bytes = (size_t)A->dim[0] * A->dim[1] * sizeof(float);
    HIPCHK(hipMemcpy(A->mat, A->mat_d, bytes, hipMemcpyDeviceToHost));
    return NMF_OK;
}

// cuda/nmf.cu:188-218 (the host copy stays raw; the EPS clamp is applied to device copies on upload)
extern "C" int nmf_read_matrix(matrix *A, const char *file) {
    if (!A || !file) return NMF_ERR_ARG;
    FILE *fp = fopen(file, "rb");
    if (!fp) { set_err("read_matrix: cannot open %s", file); return NMF_ERR_IO; }
    uint32_t rc[2];
    if (fread(rc, sizeof(uint32_t), 2, fp) != 2) { fclose(fp); set_err("read_matrix: %s: short header", file); return NMF_ERR_IO; }
    if (rc[0] == 0 || rc[1] == 0 || rc[0] > 0x7fffffffu || rc[1] > 0x7fffffffu) { fclose(fp); set_err("read_matrix: %s: bad dims", file); return NMF_ERR_IO; }
    const size_t n = (size_t)rc[0] * rc[1];
    float *buf = (float *)malloc(n * sizeof(float));
    if (!buf) { fclose(fp); return NMF_ERR_NOMEM; }
    if (fread(buf, sizeof(float), n, fp) != n) { free(buf); fclose(fp); set_err("read_matrix: %s: short data", file); return NMF_ERR_IO; }
    fclose(fp);
    A->mat = buf; A->mat_d = nullptr; A->dim[0] = (int)rc[0]; A->dim[1] = (int)rc[1];
    return NMF_OK;
}
// cuda/nmf.cu:220-259
extern "C" int nmf_write_matrix(matrix A, const char *file) {
    if (!A.mat || !file || A.dim[0] <= 0 || A.dim[1] <= 0) return NMF_ERR_ARG;
    FILE *fp = fopen(file, "wb");
    if (!fp) { set_err("write_matrix: cannot open %s", file); return NMF_ERR_IO; }
    uint32_t rc[2] = {(uint32_t)A.dim[0], (uint32_t)A.dim[1]};
    const size_t n = (size_t)A.dim[0] * A.dim[1];
    const bool ok = fwrite(rc, sizeof(uint32_t), 2, fp) == 2 && fwrite(A.mat, sizeof(float), n, fp) == n;
    fclose(fp);
    if (!ok) { set_err("write_matrix: %s: short write", file); return NMF_ERR_IO; }
    return NMF_OK;
}
