// nmf_comm.cpp -- RCCL communicator for the N-sharded W-step.  New relative to
// the reference, which is single-GPU (no NCCL/MPI anywhere in cuda/).  One process per GPU;
// xGMI all-reduce of [Z*H' ; rowsum(H)] once per iteration.
#include "nmf_comm.h"
#include "../../include/nmf_mi355x.h"

#include <dlfcn.h>
#include <cstdio>
#include <cstring>

namespace {
// minimal RCCL ABI (rccl.h): opaque comm, 128-byte unique id, enums as ints
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { ncclSuccess = 0 };
enum { ncclFloat32 = 7, ncclFloat64 = 8 };   // ncclDataType_t
enum { ncclSum = 0 };                        // ncclRedOp_t

struct Api {
    void *handle = nullptr;
    int (*GetUniqueId)(ncclUniqueId *) = nullptr;
    int (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
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
    g_api.GetUniqueId = (int (*)(ncclUniqueId *))dlsym(g_api.handle, "ncclGetUniqueId");
    g_api.CommInitRank = (int (*)(ncclComm_t *, int, ncclUniqueId, int))dlsym(g_api.handle, "ncclCommInitRank");
    g_api.CommDestroy = (int (*)(ncclComm_t))dlsym(g_api.handle, "ncclCommDestroy");
    g_api.AllReduce = (int (*)(const void *, void *, size_t, int, int, ncclComm_t, hipStream_t))dlsym(g_api.handle, "ncclAllReduce");
    g_api.GetErrorString = (const char *(*)(int))dlsym(g_api.handle, "ncclGetErrorString");
    g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.AllReduce;
    return g_api.ok;
}
}  // namespace

struct nmf_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
};

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
    const int rc = g_api.CommInitRank(&c->comm, nranks, u, rank);
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

int nmf_comm_rank(const nmf_comm *c) { return c ? c->rank : 0; }
int nmf_comm_size(const nmf_comm *c) { return c ? c->nranks : 1; }

static int allreduce(nmf_comm *c, void *buf, size_t count, int dtype, hipStream_t stream) {
    if (!c || !c->comm) return NMF_ERR_ARG;
    const int rc = g_api.AllReduce(buf, buf, count, dtype, ncclSum, c->comm, stream);
    if (rc != ncclSuccess) {
        fprintf(stderr, "nmf_comm: ncclAllReduce failed: %s\n", g_api.GetErrorString ? g_api.GetErrorString(rc) : "?");
        return NMF_ERR_COMM;
    }
    return NMF_OK;
}
int nmf_comm_allreduce_f32(nmf_comm *c, float *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat32, stream); }
int nmf_comm_allreduce_f64(nmf_comm *c, double *buf, size_t count, hipStream_t stream) { return allreduce(c, buf, count, ncclFloat64, stream); }
