// nmf_comm.h -- thin RCCL wrapper for N-sharded runs (internal).  RCCL is dlopen()ed on first
// use so that single-GPU users of libnmf_mi355x.so never load it.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>

struct nmf_comm;
// sum all-reduce in place, on `stream`; capturable into a hipGraph
int nmf_comm_allreduce_f32(nmf_comm *c, float *buf, size_t count, hipStream_t stream);
int nmf_comm_allreduce_f64(nmf_comm *c, double *buf, size_t count, hipStream_t stream);
int nmf_comm_rank(const nmf_comm *c);
int nmf_comm_size(const nmf_comm *c);
// one process, n devices (ncclCommInitAll); one host thread per rank must then drive its communicator
int nmf_comm_init_all(nmf_comm **comms, int n, const int *devices);
// n emulated ranks on the current device: host-thread rendezvous + a device-side sum in rank order (one-GPU test boxes)
int nmf_comm_create_emulated(nmf_comm **comms, int n);
// can the collective be captured into a hipGraph? (RCCL: yes; the emulated group needs its host rendezvous: no)
bool nmf_comm_capturable(const nmf_comm *c);
// called by a rank that failed outside a collective, or whose wait ran into its deadline: wakes every waiter of an emulated
// group; aborts EVERY communicator of an ncclCommInitAll group (RCCL requires all of them to be aborted: a peer inside
// ncclAllReduce keeps spinning for a rank whose own communicator alone was aborted) and nulls them, so that the pending and
// later collectives of every rank return NMF_ERR_COMM instead of waiting for ever.  Idempotent, thread-safe.
void nmf_comm_abort(nmf_comm *c);
bool nmf_comm_aborted(const nmf_comm *c);
// deadline (seconds) of nmf_comm_wait and of the emulated group's host rendezvous: NMF_COMM_TIMEOUT_S, default 30
double nmf_comm_timeout_s();
// wait for `stream` to drain, for at most `timeout_s`, watching a pinned host word the device stores a ticket into (no HIP call
// while waiting: see nmf_comm.cpp).  On expiry (a rank that never arrived at a collective leaves
// its peers' all-reduce kernels spinning) or when another rank has aborted the group: abort the group, give the stream a
// few seconds to drain its now-aborted collective, and return NMF_ERR_COMM.  c == nullptr: plain hipStreamSynchronize.
int nmf_comm_wait(nmf_comm *c, hipStream_t stream, double timeout_s, const char *what);

// Liveness of a rank's host thread: bumped on entry to and return from every collective call and on every poll of
// nmf_comm_wait.  A rank whose value stops changing is blocked inside a host call (RCCL connects its transports inside the
// first collective's enqueue and waits there for its peers): only another thread can help it, by aborting the group --
// nmf_update_div_multi's calling thread watches the ranks for that (nmf_multi.cpp).
long nmf_comm_heartbeat(const nmf_comm *c);

// the calling thread's nmf_last_error() text (rank threads hand their message to the thread that called update_div_ex)
void nmf_internal_set_error(const char *msg);

// dst[i] = src[0][i] + src[1][i] + ... + src[n - 1][i] in rank order (the same bits on every rank), f32 or f64: the emulated
// group's reduction, launched on `stream` (nmf_kernels.hip).  n <= NMF_EMU_MAX_RANKS.
#define NMF_EMU_MAX_RANKS 8
hipError_t nmf_emu_sum_launch(const void *const *src, int n, void *dst, size_t count, bool f64, hipStream_t stream);
// *word_host_mapped = value (system-scope release store) by one thread on `stream`: the completion ticket of nmf_comm_wait (nmf_kernels.hip)
hipError_t nmf_flag_store_launch(unsigned *word_host_mapped, unsigned value, hipStream_t stream);
// how often a wait's closing hipStreamSynchronize was refused because of a capture elsewhere in the process (expected: never)
// (declared in include/nmf_mi355x.h: nmf_comm_capture_refusals)
