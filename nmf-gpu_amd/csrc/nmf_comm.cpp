// nmf_comm.cpp -- RCCL communicator for the N-sharded W-step.  New relative to
// the reference, which is single-GPU (no NCCL/MPI anywhere in cuda/).  One process per GPU;
// xGMI all-reduce of [Z*H' ; rowsum(H)] once per iteration.
#include "nmf_comm.h"
#include "../../include/nmf_mi355x.h"

#include <dlfcn.h>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <memory>
#include <mutex>
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
    bool ok = false;
};
Api g_api;

bool load_api() {
    if (g_api.ok) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *n : names) {
        g_api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_api.handle) break;
    }
    if (!g_api.handle) { fprintf(stderr, "nmf_comm: cannot dlopen librccl: %s\n", dlerror()); return false; }
    g_api.GetUniqueId = (decltype(&ncclGetUniqueId))dlsym(g_api.handle, "ncclGetUniqueId");
    g_api.CommInitRank = (decltype(&ncclCommInitRank))dlsym(g_api.handle, "ncclCommInitRank");
    g_api.CommInitAll = (decltype(&ncclCommInitAll))dlsym(g_api.handle, "ncclCommInitAll");
    g_api.CommDestroy = (decltype(&ncclCommDestroy))dlsym(g_api.handle, "ncclCommDestroy");
    g_api.AllReduce = (decltype(&ncclAllReduce))dlsym(g_api.handle, "ncclAllReduce");
    g_api.GetErrorString = (decltype(&ncclGetErrorString))dlsym(g_api.handle, "ncclGetErrorString");
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
    bool rendezvous() {              // false: the group was aborted
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return false;
        const unsigned long gen = generation;
        if (++arrived == n) { arrived = 0; ++generation; cv.notify_all(); }
        else cv.wait(lk, [&] { return generation != gen || aborted; });
        return !aborted;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

struct nmf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    std::shared_ptr<EmuGroup> emu;   // set: a same-device emulated group instead of an RCCL communicator
};

struct EmuPtrs { const void *p[kMaxEmu]; };
template <typename T>
__global__ __launch_bounds__(256) void emu_sum_kernel(EmuPtrs src, int n, T *__restrict__ dst, size_t count) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) {
        T s = reinterpret_cast<const T *>(src.p[0])[i];
        for (int r = 1; r < n; ++r) s += reinterpret_cast<const T *>(src.p[r])[i];   // rank order: the same bits on every rank
        dst[i] = s;
    }
}

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
    const ncclResult_t rc = g_api.CommInitRank(&c->comm, nranks, u, rank);
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclCommInitRank failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        delete c;
        return NMF_ERR_COMM;
    }
    *out = c;
    return NMF_OK;
}

extern "C" void nmf_comm_destroy(nmf_comm *c) {
    if (!c) return;
    if (c->comm && g_api.ok) g_api.CommDestroy(c->comm);
    delete c;
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
    for (int i = 0; i < n; ++i) {
        out[i] = new nmf_comm();
        out[i]->comm = cs[(size_t)i]; out[i]->rank = i; out[i]->nranks = n;
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
    }
    return NMF_OK;
}
bool nmf_comm_capturable(const nmf_comm *c) { return c && !c->emu; }
// a rank that has failed outside a collective tells the group, so that the others do not wait for it for ever
void nmf_comm_abort(nmf_comm *c) {
    if (!c) return;
    if (c->emu) { c->emu->abort(); return; }
    if (c->comm && g_api.ok) {
        auto fn = (decltype(&ncclCommAbort))dlsym(g_api.handle, "ncclCommAbort");
        if (fn) { (void)fn(c->comm); c->comm = nullptr; }
    }
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
    if (!g.rendezvous()) return NMF_ERR_COMM;          // every operand is enqueued and published
    EmuPtrs ptrs;
    for (int h = 0; h < n; ++h) {
        ptrs.p[h] = g.buf[h];
        if (h != r && hipStreamWaitEvent(stream, g.ev1[h], 0) != hipSuccess) return NMF_ERR_HIP;
    }
    size_t grid = (count + 255) / 256;
    if (grid > 1024) grid = 1024;
    hipLaunchKernelGGL(emu_sum_kernel<T>, dim3((unsigned)grid), dim3(256), 0, stream, ptrs, n, (T *)g.tmp[r], count);
    if (hipEventRecord(g.ev2[r], stream) != hipSuccess) return NMF_ERR_HIP;
    if (!g.rendezvous()) return NMF_ERR_COMM;          // every rank has read every operand ...
    for (int h = 0; h < n; ++h)
        if (h != r && hipStreamWaitEvent(stream, g.ev2[h], 0) != hipSuccess) return NMF_ERR_HIP;
    if (hipMemcpyAsync(buf, g.tmp[r], bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return NMF_ERR_HIP;   // ... before any is overwritten
    return NMF_OK;
}

int nmf_comm_rank(const nmf_comm *c) { return c ? c->rank : 0; }
int nmf_comm_size(const nmf_comm *c) { return c ? c->nranks : 1; }

static int allreduce(nmf_comm *c, void *buf, size_t count, ncclDataType_t dtype, hipStream_t stream) {
    if (c && c->emu) return dtype == ncclFloat32 ? emu_allreduce(c, (float *)buf, count, stream) : emu_allreduce(c, (double *)buf, count, stream);
    if (!c || !c->comm) return NMF_ERR_ARG;
    const ncclResult_t rc = g_api.AllReduce(buf, buf, count, dtype, ncclSum, c->comm, stream);
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclAllReduce failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        return NMF_ERR_COMM;
    }
    return NMF_OK;
}
int nmf_comm_allreduce_f32(nmf_comm *c, float *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat32, stream); }
int nmf_comm_allreduce_f64(nmf_comm *c, double *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat64, stream); }
