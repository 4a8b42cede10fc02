// nmf_comm.cpp -- RCCL communicator for the N-sharded W-step.  New relative to
// the reference, which is single-GPU (no NCCL/MPI anywhere in cuda/).  One process per GPU;
// xGMI all-reduce of [Z*H' ; rowsum(H)] once per iteration.
#include "nmf_comm.h"
#include "../../include/nmf_mi355x.h"

#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

// Types, enums and prototypes come from the RCCL header (compile time only: the library itself is dlopen()ed on first use,
// so single-GPU users of libnmf_mi355x.so never load it); the function-pointer table below is checked against the header's
// prototypes by the decltype casts in load_api().
#include <rccl/rccl.h>

namespace {
struct Api {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    int version = 0;            // ncclGetVersion of the loaded library
    char path[400] = "";        // where it was loaded from (dladdr)
    bool ok = false;
};
Api g_api;
std::mutex g_api_mu;

// The function table is typed by <rccl/rccl.h> (NCCL_VERSION_CODE of the ROCm this library was compiled against).  Which shared
// object it binds to: one the process has ALREADY mapped if there is one (under PyTorch that is torch's bundled build; two RCCLs in
// one process would each set up their own transports), else the ROCm installation's own -- the one the header belongs to -- and
// only then whatever the loader finds under the bare name.  The entry points used here (all-reduce, init, destroy, abort, 128-byte
// unique id) have kept their signatures through NCCL 2.x; a different MAJOR version is refused, an older minor than the header's
// is reported once on stderr, the loaded version and path are available through nmf_comm_library_info.
bool load_api() {
    std::lock_guard<std::mutex> lock(g_api_mu);
    if (g_api.ok) return true;
    for (const char *n : {"librccl.so.1", "librccl.so"}) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);   // already in the process?
        if (g_api.handle) break;
    }
    if (!g_api.handle) {
        char rocm[512];
        const char *root = getenv("ROCM_PATH");
        snprintf(rocm, sizeof rocm, "%s/lib/librccl.so.1", (root && root[0]) ? root : "/opt/rocm");
        const char *names[] = {rocm, "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (g_api.handle) break;
        }
    }
    if (!g_api.handle) { fprintf(stderr, "nmf_comm: cannot dlopen librccl: %s\n", dlerror()); return false; }
    g_api.GetUniqueId = (decltype(&ncclGetUniqueId))dlsym(g_api.handle, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(&ncclCommInitRank))dlsym(g_api.handle, "ncclCommInitRank");
    g_api.CommInitAll = (decltype(&ncclCommInitAll))dlsym(g_api.handle, "ncclCommInitAll");
    g_api.CommDestroy = (decltype(&ncclCommDestroy))dlsym(g_api.handle, "ncclCommDestroy");
    g_api.AllReduce = (decltype(&ncclAllReduce))dlsym(g_api.handle, "ncclAllReduce");
    g_api.GetErrorString = (decltype(&ncclGetErrorString))dlsym(g_api.handle, "ncclGetErrorString");
    g_api.CommAbort = (decltype(&ncclCommAbort))dlsym(g_api.handle, "ncclCommAbort");
    g_api.GetVersion = (decltype(&ncclGetVersion))dlsym(g_api.handle, "ncclGetVersion");
    if (g_api.GetVersion) { int v = 0; if (g_api.GetVersion(&v) == ncclSuccess) g_api.version = v; }
    Dl_info di;
    if (g_api.AllReduce && dladdr((void *)g_api.AllReduce, &di) && di.dli_fname) snprintf(g_api.path, sizeof g_api.path, "%s", di.dli_fname);
    const int major = g_api.version >= 10000 ? g_api.version / 10000 : g_api.version / 1000;   // NCCL_VERSION(X,Y,Z): X*10000 + Y*100 + Z since 2.9
    if (g_api.version && major != NCCL_MAJOR) {
        fprintf(stderr, "nmf_comm: %s is RCCL/NCCL %d, this library was built against the %d.x interface (rccl.h %d): refusing it\n",
                g_api.path, g_api.version, NCCL_MAJOR, NCCL_VERSION_CODE);
        return false;
    }
    if (g_api.version && g_api.version < NCCL_VERSION_CODE)
        fprintf(stderr, "nmf_comm: note: %s is RCCL %d.%d.%d, older than the rccl.h this library was built against (%d.%d.%d); the entry points used "
                        "are unchanged across 2.x\n", g_api.path, g_api.version / 10000, (g_api.version / 100) % 100, g_api.version % 100, NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce;
    return g_api.ok;
}
}  // namespace

// A same-device group: `n` shards of one problem on ONE GPU, each driven by its own host thread and stream, with the
// all-reduce done by a device kernel behind two host rendezvous.  It exists so that the multi-device driver (nmf_multi.cpp:
// threads, column shards, W broadcast, per-rank loop, H gather) can be run and checked on a one-GPU box; the arithmetic is
// the all-reduce's contract, sum over ranks in rank order, so every rank ends up with the same bits.
constexpr int kMaxEmu = 8;
struct EmuGroup {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    unsigned long generation = 0;
    bool aborted = false;            // a rank has failed: nobody waits for it any more
    const void *buf[kMaxEmu] = {};
    hipEvent_t ev1[kMaxEmu] = {}, ev2[kMaxEmu] = {};
    void *tmp[kMaxEmu] = {};
    size_t tmp_bytes[kMaxEmu] = {};
    ~EmuGroup() {
        for (int i = 0; i < n; ++i) {
            if (ev1[i]) (void)hipEventDestroy(ev1[i]);
            if (ev2[i]) (void)hipEventDestroy(ev2[i]);
            if (tmp[i]) (void)hipFree(tmp[i]);
        }
    }
    bool rendezvous(double timeout_s) {   // false: the group was aborted, or a rank did not arrive within the deadline (aborts it)
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const unsigned long gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else if (getenv("NMF_FAULT_BLOCKING_COLLECTIVE")) cv.wait(lk, [&] { return generation != gen || aborted; });   // fault injection: no deadline
        // of its own, like a host call that blocks inside RCCL: only the watchdog of nmf_update_div_multi ends it
        else if (!cv.wait_for(lk, std::chrono::duration<double>(timeout_s), [&] { return generation != gen || aborted; })) {
            fprintf(stderr, "nmf_comm: a rank of the emulated group did not reach its collective within %.1f s; aborting the group\n", timeout_s);
            aborted = true;
            cv.notify_all();
        }
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
    bool is_aborted() { std::lock_guard<std::mutex> lk(mu); return aborted; }
};

// The communicators of one ncclCommInitAll call: aborting one means aborting all (see nmf_comm_abort)
struct RcclGroup {
    std::mutex mu;
    std::vector<nmf_comm *> members;
    std::atomic<bool> aborted{false};
};

struct nmf_comm {
    // The RCCL handle.  ncclCommAbort frees it, and the thread that aborts a group is usually not the thread that uses this
    // communicator: a user takes call_mu, re-checks `aborted`, loads the handle and keeps call_mu until its RCCL call has
    // returned; the aborter marks every member aborted first, takes the handle out (exchange with nullptr) and then waits for
    // call_mu -- briefly: a rank between its check and the end of an ordinary enqueue finishes first, so the handle is never
    // freed under it; a rank that stays BLOCKED inside the call (RCCL waiting for a dead peer) is exactly what ncclCommAbort
    // exists to wake, and gets it after the grace period.
    std::atomic<ncclComm_t> comm{nullptr};
    std::timed_mutex call_mu;
    int rank = 0, nranks = 1;
    std::shared_ptr<EmuGroup> emu;   // set: a same-device emulated group instead of an RCCL communicator
    std::shared_ptr<RcclGroup> grp;  // set: one of the communicators of an ncclCommInitAll group
    std::atomic<bool> aborted{false};
    std::atomic<long> beat{0};       // nmf_comm_heartbeat
    // fault injection for the tests (NMF_FAULT_ALLREDUCE=<rank>:<call>): that f32 all-reduce call of that rank fails
    long calls = 0, fail_at = 0;
    // nmf_comm_wait: a pinned host word the device stores a ticket into behind the awaited work (allocated at the first wait)
    unsigned *ticket_word = nullptr;
    unsigned ticket = 0;
    ~nmf_comm() { if (ticket_word) (void)hipHostFree(ticket_word); }
};

double nmf_comm_timeout_s() {
    const char *e = getenv("NMF_COMM_TIMEOUT_S");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 30.0;
}
static void arm_fault(nmf_comm *c) {
    const char *e = getenv("NMF_FAULT_ALLREDUCE");
    int r = -1; long at = 0;
    if (e && sscanf(e, "%d:%ld", &r, &at) == 2 && r == c->rank) c->fail_at = at;
}

// the device-side sum of the emulated group lives with the other kernels (nmf_kernels.hip): this file is host code only, and the
// ThreadSanitizer build of it (tests/cpu_sanitize/) is a plain C++ compile with no device pass
static_assert(kMaxEmu == NMF_EMU_MAX_RANKS, "EmuGroup and the emulation kernel agree on the rank limit");

extern "C" int nmf_comm_get_unique_id(unsigned char id[NMF_COMM_ID_BYTES]) {
    if (!id) return NMF_ERR_ARG;
    if (!load_api()) return NMF_ERR_COMM;
    ncclUniqueId u;
    if (g_api.GetUniqueId(&u) != ncclSuccess) return NMF_ERR_COMM;
    static_assert(sizeof(u) == NMF_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id, &u, sizeof u);
    return NMF_OK;
}

extern "C" int nmf_comm_init_rank(nmf_comm **out, const unsigned char id[NMF_COMM_ID_BYTES], int rank, int nranks) {
    if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) return NMF_ERR_ARG;
    if (!load_api()) return NMF_ERR_COMM;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    nmf_comm *c = new nmf_comm();
    c->rank = rank; c->nranks = nranks;
    ncclComm_t h = nullptr;
    const ncclResult_t rc = g_api.CommInitRank(&h, nranks, u, rank);
    c->comm = h;
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclCommInitRank failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        delete c;
        return NMF_ERR_COMM;
    }
    arm_fault(c);
    *out = c;
    return NMF_OK;
}

extern "C" void nmf_comm_destroy(nmf_comm *c) {
    if (!c) return;
    if (c->grp) {
        std::lock_guard<std::mutex> lk(c->grp->mu);
        for (auto &m : c->grp->members) if (m == c) m = nullptr;
        const ncclComm_t h = c->comm.exchange(nullptr);   // nullptr after an abort
        if (h && g_api.ok) g_api.CommDestroy(h);
    } else {
        const ncclComm_t h = c->comm.exchange(nullptr);
        if (h && g_api.ok) g_api.CommDestroy(h);
    }
    delete c;
}

// One small all-reduce through the communicator, waited for with a deadline: every rank contributes 1.0 in eight floats and must read
// back the number of ranks.  The first collective of a communicator is where RCCL brings up its transports and where a peer that never
// arrives shows: a caller that can still choose another all-reduce (bench.py: torch.distributed) probes before it commits.  Collective:
// every rank of the communicator must call it.  On expiry the communicator (group) is aborted, as with every other wait.
extern "C" int nmf_comm_probe(nmf_comm *c, double timeout_s) {
    if (!c) return NMF_ERR_ARG;
    if (timeout_s <= 0.0) timeout_s = nmf_comm_timeout_s();
    float *d = nullptr;
    hipStream_t st = nullptr;
    if (hipMalloc((void **)&d, 8 * sizeof(float)) != hipSuccess) return NMF_ERR_NOMEM;
    int rc = NMF_OK;
    const float ones[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
    float back[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { (void)hipFree(d); return NMF_ERR_HIP; }
    if (hipMemcpyAsync(d, ones, sizeof ones, hipMemcpyHostToDevice, st) != hipSuccess) rc = NMF_ERR_HIP;
    if (rc == NMF_OK) rc = nmf_comm_allreduce_f32(c, d, 8, st);
    if (rc == NMF_OK) rc = nmf_comm_wait(c, st, timeout_s, "the probe all-reduce");
    if (rc == NMF_OK && hipMemcpyAsync(back, d, sizeof back, hipMemcpyDeviceToHost, st) != hipSuccess) rc = NMF_ERR_HIP;
    if (rc == NMF_OK && hipStreamSynchronize(st) != hipSuccess) rc = NMF_ERR_HIP;
    if (rc == NMF_OK)
        for (float v : back)
            if (v != (float)c->nranks) { fprintf(stderr, "nmf_comm: rank %d: the probe all-reduce returned %g, expected %d\n", c->rank, (double)v, c->nranks); rc = NMF_ERR_COMM; break; }
    (void)hipStreamDestroy(st);
    (void)hipFree(d);
    return rc;
}

extern "C" int nmf_comm_library_info(char *buf, int buflen) {
    if (!buf || buflen <= 0) return NMF_ERR_ARG;
    if (!load_api()) { snprintf(buf, (size_t)buflen, "RCCL not loadable"); return NMF_ERR_COMM; }
    const int v = g_api.version;
    snprintf(buf, (size_t)buflen, "RCCL %d.%d.%d (%s); built against rccl.h %d.%d.%d", v / 10000, (v / 100) % 100, v % 100, g_api.path[0] ? g_api.path : "?",
             NCCL_MAJOR, NCCL_MINOR, NCCL_PATCH);
    return NMF_OK;
}

// ncclCommInitAll: one process drives n devices (devices[i] = HIP ordinal of rank i), one host thread per rank afterwards
int nmf_comm_init_all(nmf_comm **out, int n, const int *devices) {
    if (!out || n < 1 || !devices) return NMF_ERR_ARG;
    if (!load_api() || !g_api.CommInitAll) return NMF_ERR_COMM;
    std::vector<ncclComm_t> cs((size_t)n, nullptr);
    const ncclResult_t rc = g_api.CommInitAll(cs.data(), n, devices);
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclCommInitAll failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        return NMF_ERR_COMM;
    }
    auto grp = std::make_shared<RcclGroup>();
    for (int i = 0; i < n; ++i) {
        out[i] = new nmf_comm();
        out[i]->comm = cs[(size_t)i]; out[i]->rank = i; out[i]->nranks = n; out[i]->grp = grp;
        grp->members.push_back(out[i]);
        arm_fault(out[i]);
    }
    return NMF_OK;
}

// n emulated ranks on the current device (see EmuGroup)
int nmf_comm_create_emulated(nmf_comm **out, int n) {
    if (!out || n < 1 || n > kMaxEmu) return NMF_ERR_ARG;
    auto g = std::make_shared<EmuGroup>();
    g->n = n;
    for (int i = 0; i < n; ++i) {
        if (hipEventCreateWithFlags(&g->ev1[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&g->ev2[i], hipEventDisableTiming) != hipSuccess)
            return NMF_ERR_HIP;
    }
    for (int i = 0; i < n; ++i) {
        out[i] = new nmf_comm();
        out[i]->rank = i; out[i]->nranks = n; out[i]->emu = g;
        arm_fault(out[i]);
    }
    return NMF_OK;
}
bool nmf_comm_capturable(const nmf_comm *c) { return c && !c->emu; }
// a rank that has failed outside a collective (or timed out inside one) tells the group, so that nobody waits for ever
void nmf_comm_abort(nmf_comm *c) {
    if (!c) return;
    c->aborted = true;
    if (c->emu) { c->emu->abort(); return; }
    if (!g_api.ok || !g_api.CommAbort) return;
    // no new RCCL call starts on `m` once m->aborted is set (allreduce_impl checks it under call_mu); one that is under way gets a
    // grace period to return before its handle is freed (see nmf_comm::comm)
    auto abort_one = [](nmf_comm *m) {
        const ncclComm_t h = m->comm.exchange(nullptr);
        if (!h) return;
        std::unique_lock<std::timed_mutex> quiet(m->call_mu, std::defer_lock);
        (void)quiet.try_lock_for(std::chrono::milliseconds(200));
        (void)g_api.CommAbort(h);
    };
    if (c->grp) {   // every communicator of the group: the peers' kernels poll their OWN communicator's abort flag
        std::lock_guard<std::mutex> lk(c->grp->mu);
        c->grp->aborted = true;
        for (nmf_comm *m : c->grp->members) if (m) m->aborted = true;
        for (nmf_comm *m : c->grp->members) if (m) abort_one(m);
        return;
    }
    abort_one(c);
}
long nmf_comm_heartbeat(const nmf_comm *c) { return c ? c->beat.load(std::memory_order_relaxed) : 0; }
bool nmf_comm_aborted(const nmf_comm *c) {
    if (!c) return false;
    if (c->aborted) return true;
    if (c->emu) return c->emu->is_aborted();
    return c->grp ? c->grp->aborted.load() : false;
}

// Waiting for a stream with a deadline, WITHOUT polling it through the HIP API.
//
// Until round 5 this loop called hipStreamQuery every few microseconds, and once -- in the rehearsal that runs several host threads
// of one process side by side, each capturing hipGraphs with RCCL all-reduces inside -- the query answered with an error on the
// caller's own, healthy stream.  What this HIP build (libamdhip64.so.7.2.70200) can answer from hipStreamQuery, read off its code:
// hipErrorNotReady / hipSuccess; the process-wide sticky status every API entry starts with (a device fault: never transient);
// hipErrorInvalidHandle from the look-up of the stream in the runtime's set of live streams; and hipErrorStreamCaptureUnsupported
// from the capture rule (a stream that is not capturing is refused while a capture this thread must respect is under way).  The last
// two do not describe the polled stream: they describe the runtime's stream set and capture lists at that instant, which the OTHER
// threads were changing (hipStreamBeginCapture / EndCapture, RCCL forking its internal stream into the capture on ncclAllReduce,
// hipGraphInstantiate, pooled-stream creation).  A poll that asks thousands of times per wait is the call most exposed to them.
// So the wait no longer asks the runtime anything while it waits: a one-thread kernel behind the awaited work stores a ticket into
// a pinned host word, the host watches that word, and ONE hipStreamSynchronize -- on a stream that has already drained -- collects
// the stream's status, which is taken at face value (with one named exception, below).  Captures are also serialised process-wide and,
// inside one sharded call, finished on every rank before any rank replays (nmf_host.cpp: capture_graph; nmf_multi.cpp: the set-up gate).
namespace {
std::atomic<long> g_capture_refusals{0};
}
extern "C" long nmf_comm_capture_refusals(void) { return g_capture_refusals.load(); }

static int wait_ticket_init(nmf_comm *c) {
    if (c->ticket_word) return NMF_OK;
    void *p = nullptr;
    if (hipHostMalloc(&p, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { (void)hipGetLastError(); return NMF_ERR_NOMEM; }
    *(volatile unsigned *)p = 0u;
    c->ticket_word = (unsigned *)p;
    return NMF_OK;
}

int nmf_comm_wait(nmf_comm *c, hipStream_t stream, double timeout_s, const char *what) {
    if (!c) return hipStreamSynchronize(stream) == hipSuccess ? NMF_OK : NMF_ERR_HIP;
    if (wait_ticket_init(c) != NMF_OK) return NMF_ERR_NOMEM;
    const unsigned ticket = ++c->ticket;
    {
        const hipError_t e = nmf_flag_store_launch(c->ticket_word, ticket, stream);
        if (e != hipSuccess) { fprintf(stderr, "nmf_comm: rank %d: cannot enqueue the completion ticket behind %s: %s\n", c->rank, what ? what : "a collective", hipGetErrorName(e)); return NMF_ERR_HIP; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    bool expired = false;
    for (int spins = 0;; ++spins) {
        c->beat.fetch_add(1, std::memory_order_relaxed);
        if (__atomic_load_n(c->ticket_word, __ATOMIC_ACQUIRE) == ticket) break;   // the device wrote it: everything in front of it has finished
        const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (!expired && (el > timeout_s || nmf_comm_aborted(c))) {
            if (el > timeout_s) fprintf(stderr, "nmf_comm: rank %d: %s did not complete within %.1f s; aborting the communicator group\n", c->rank, what ? what : "a collective", timeout_s);
            nmf_comm_abort(c);
            expired = true;
        }
        if (expired && el > timeout_s + 10.0) return NMF_ERR_COMM;   // the aborted collective should have left the stream by now
        if (spins < 2000) std::this_thread::yield(); else std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    // The stream has drained; this returns at once and its answer is the stream's real status.  hipErrorStreamCaptureUnsupported alone is
    // asked again (it reports a capture elsewhere, not this stream), at most three times, and counted: nmf_comm_capture_refusals() must
    // stay 0 in the tests.  Anything else is an error of this run and ends it, by name.
    for (int attempt = 0;; ++attempt) {
        const hipError_t e = hipStreamSynchronize(stream);
        if (e == hipSuccess) break;
        fprintf(stderr, "nmf_comm: rank %d: hipStreamSynchronize after %s: %s (%s)\n", c->rank, what ? what : "a collective", hipGetErrorName(e), hipGetErrorString(e));
        if (e == hipErrorStreamCaptureUnsupported && attempt < 3) { g_capture_refusals.fetch_add(1); std::this_thread::sleep_for(std::chrono::milliseconds(1)); continue; }
        return NMF_ERR_HIP;
    }
    return (expired || nmf_comm_aborted(c)) ? NMF_ERR_COMM : NMF_OK;
}

template <typename T>
static int emu_allreduce(nmf_comm *c, T *buf, size_t count, hipStream_t stream) {
    EmuGroup &g = *c->emu;
    const int r = c->rank, n = g.n;
    const size_t bytes = count * sizeof(T);
    if (g.tmp_bytes[r] < bytes) {
        if (g.tmp[r]) (void)hipFree(g.tmp[r]);
        g.tmp[r] = nullptr; g.tmp_bytes[r] = 0;
        if (hipMalloc(&g.tmp[r], bytes) != hipSuccess) return NMF_ERR_NOMEM;
        g.tmp_bytes[r] = bytes;
    }
    g.buf[r] = buf;
    if (hipEventRecord(g.ev1[r], stream) != hipSuccess) return NMF_ERR_HIP;
    if (!g.rendezvous(nmf_comm_timeout_s())) return NMF_ERR_COMM;          // every operand is enqueued and published
    const void *ptrs[kMaxEmu] = {};
    for (int h = 0; h < n; ++h) {
        ptrs[h] = g.buf[h];
        if (h != r && hipStreamWaitEvent(stream, g.ev1[h], 0) != hipSuccess) return NMF_ERR_HIP;
    }
    if (nmf_emu_sum_launch(ptrs, n, g.tmp[r], count, sizeof(T) == 8, stream) != hipSuccess) return NMF_ERR_HIP;
    if (hipEventRecord(g.ev2[r], stream) != hipSuccess) return NMF_ERR_HIP;
    if (!g.rendezvous(nmf_comm_timeout_s())) return NMF_ERR_COMM;          // every rank has read every operand ...
    for (int h = 0; h < n; ++h)
        if (h != r && hipStreamWaitEvent(stream, g.ev2[h], 0) != hipSuccess) return NMF_ERR_HIP;
    if (hipMemcpyAsync(buf, g.tmp[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return NMF_ERR_HIP;   // ... before any is overwritten
    return NMF_OK;
}

int nmf_comm_rank(const nmf_comm *c) { return c ? c->rank : 0; }
int nmf_comm_size(const nmf_comm *c) { return c ? c->nranks : 1; }

static int allreduce_impl(nmf_comm *c, void *buf, size_t count, ncclDataType_t dtype, hipStream_t stream);
static int allreduce(nmf_comm *c, void *buf, size_t count, ncclDataType_t dtype, hipStream_t stream) {
    if (c) c->beat.fetch_add(1, std::memory_order_relaxed);
    const int st = allreduce_impl(c, buf, count, dtype, stream);
    if (c) c->beat.fetch_add(1, std::memory_order_relaxed);
    return st;
}
static int allreduce_impl(nmf_comm *c, void *buf, size_t count, ncclDataType_t dtype, hipStream_t stream) {
    if (c && dtype == ncclFloat32 && c->fail_at > 0 && ++c->calls == c->fail_at) {
        fprintf(stderr, "nmf_comm: rank %d: injected all-reduce failure (NMF_FAULT_ALLREDUCE)\n", c->rank);
        return NMF_ERR_COMM;
    }
    if (c && nmf_comm_aborted(c)) return NMF_ERR_COMM;
    if (c && c->emu) return dtype == ncclFloat32 ? emu_allreduce(c, (float *)buf, count, stream) : emu_allreduce(c, (double *)buf, count, stream);
    if (!c) return NMF_ERR_ARG;
    std::lock_guard<std::timed_mutex> in_call(c->call_mu);   // nmf_comm_abort frees the handle only after this call has returned (or stayed blocked)
    if (nmf_comm_aborted(c)) return NMF_ERR_COMM;
    const ncclComm_t h = c->comm.load();
    if (!h) return NMF_ERR_COMM;
    const ncclResult_t rc = g_api.AllReduce(buf, buf, count, dtype, ncclSum, h, stream);
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclAllReduce failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        return NMF_ERR_COMM;
    }
    return NMF_OK;
}
int nmf_comm_allreduce_f32(nmf_comm *c, float *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat32, stream); }
int nmf_comm_allreduce_f64(nmf_comm *c, double *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat64, stream); }
