/* A stand-in librccl.so.1 for tests/test_comm_race.py (CPU, ThreadSanitizer): our own test double, not RCCL.  It keeps a
 * heap object per communicator and touches it the way a real library would -- ncclAllReduce reads and writes the object
 * (with a small delay inside the call), ncclCommAbort and ncclCommDestroy free it -- so that a use of a communicator handle
 * after (or while) it is freed shows up under ThreadSanitizer as a data race / heap-use-after-free on that object. */
#include <rccl/rccl.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

struct ncclComm { volatile long ops; volatile int alive; char pad[240]; };

ncclResult_t ncclGetVersion(int *v) { *v = NCCL_VERSION_CODE; return ncclSuccess; }
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "ok" : "fake rccl error"; }
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof *id); return ncclSuccess; }
static ncclComm_t make(void) { ncclComm_t c = (ncclComm_t)calloc(1, sizeof(struct ncclComm)); c->alive = 1; return c; }
ncclResult_t ncclCommInitRank(ncclComm_t *c, int n, ncclUniqueId id, int rank) { (void)n; (void)id; (void)rank; *c = make(); return ncclSuccess; }
ncclResult_t ncclCommInitAll(ncclComm_t *c, int n, const int *devs) { (void)devs; for (int i = 0; i < n; ++i) c[i] = make(); return ncclSuccess; }
ncclResult_t ncclCommDestroy(ncclComm_t c) { c->alive = 0; free(c); return ncclSuccess; }
ncclResult_t ncclCommAbort(ncclComm_t c) { c->alive = 0; free(c); return ncclSuccess; }
ncclResult_t ncclAllReduce(const void *s, void *r, size_t n, ncclDataType_t t, ncclRedOp_t op, ncclComm_t c, hipStream_t st) {
    (void)s; (void)r; (void)n; (void)t; (void)op; (void)st;
    if (!c->alive) return ncclInvalidUsage;
    c->ops++;
    if ((c->ops & 7) == 0) usleep(20);      /* stay inside the call for a while now and then */
    c->ops++;
    return c->alive ? ncclSuccess : ncclInvalidUsage;
}
