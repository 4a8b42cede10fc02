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
