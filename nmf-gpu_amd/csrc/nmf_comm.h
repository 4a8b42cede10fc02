// nmf_comm.h -- thin RCCL wrapper for N-sharded runs (internal).  RCCL is dlopen()ed on first
// use so that single-GPU users of libnmf_mi355x.so never load it.
#pragma once
#include <hip/hip_runtime.h>
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
// called by a rank that failed outside a collective: wakes (emulated group) or aborts (RCCL) the communicator so that the other
// ranks' pending collectives return an error instead of waiting for ever
void nmf_comm_abort(nmf_comm *c);
