"""N-sharded update_div: one process per GPU, rank g owns the column block J_g of X
and H plus a full copy of W.  The H half-step is local; the W half-step needs one sum all-reduce
per iteration of the (M*K + K)-float buffer [Z_g * H_g' ; rowsum(H_g)], after which every rank
applies the same update and W stays replicated bit for bit.  New relative to the reference
(single GPU, no NCCL/MPI anywhere in cuda/).

This module is host logic only: it drives any *backend* exposing the half-step protocol

    update_h()                      local H half-step                      (cuda/nmf.cu:118-146)
    w_partial() -> buffer           leaves the partial buffer ready        (first half of 148-176)
    w_apply()                       W *= psum / max(hsum, EPS)             (second half)
    check_local() -> (kl, sum|x-y|, sum|x|)

and an ``allreduce_sum(buffer)`` callable.  The GPU backend is :class:`GpuShard` (the HIP solver
with a torch tensor as the all-reduce operand); tests drive the same loop with a CPU backend
over gloo.
"""
from __future__ import annotations

from typing import Callable, List, Tuple


def column_shards(N: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous column blocks [start, stop) per rank; the first N % world ranks get one extra
    column.  Every rank must own at least one column."""
    if world < 1 or N < world:
        raise ValueError(f"cannot shard {N} columns over {world} ranks")
    base, extra = divmod(N, world)
    out, start = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((start, start + cnt))
        start += cnt
    return out


def worth_sharding(M: int, N: int, K: int, world: int) -> bool:
    """north_star: shard 'only where N is large enough to amortise the all-reduce'.  Per-GPU compute
    per iteration is 8*M*(N/world)*K flop at ~1e14 flop/s; the all-reduce moves 4*(M*K+K) bytes at
    ~1e11 B/s per xGMI link plus ~20 us of latency.  Shard when compute >= 4x that."""
    if world <= 1:
        return False
    compute_s = 8.0 * M * (N / world) * K / 1.0e14
    comm_s = 2.0 * 4.0 * (M * K + K) / 1.0e11 + 20e-6
    return compute_s >= 4.0 * comm_s


def negotiate_comm(dist, torch, rank: int, world: int, unique_id: Callable, make_comm: Callable, device: str = "cuda"):
    """Set up one in-library communicator per rank -- rank 0 makes the 128-byte id (``unique_id()``), torch broadcasts it,
    every rank calls ``make_comm(id_bytes, rank, world)`` -- such that EVERY rank runs the same sequence of torch collectives
    whether or not something fails: rank 0 always broadcasts (a zeroed id when it could not make one) and an ok flag is
    MIN-reduced after each stage (id, communicator, a probe all-reduce through it), so a rank that fails early never leaves the others waiting in a different collective
    (round-2 advisor: a rank raising before the broadcast deadlocked the rest until the process-group time-out).
    Returns (comm, "") on every rank, or (None, reason) on every rank; a communicator made by a rank whose peers failed is
    closed before returning."""
    def all_ok(ok):
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    why, comm, ok = "", None, True
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        try:
            uid = torch.frombuffer(bytearray(unique_id()), dtype=torch.uint8).clone()
        except Exception as e:      # no librccl, wrong major version, ...
            ok, why = False, str(e)
    uid = uid.to(device)
    dist.broadcast(uid, 0)
    ok = all_ok(ok)
    if ok:
        try:
            comm = make_comm(bytes(uid.cpu().numpy().tobytes()), rank, world)
        except Exception as e:      # ncclCommInitRank failure (e.g. two ranks on one device)
            ok, why = False, str(e)
        ok = all_ok(ok)
    if ok and hasattr(comm, "probe"):
        # the communicator's FIRST collective, with a deadline (NMF_COMM_TIMEOUT_S): transports come up here and a peer that cannot
        # be reached shows here -- while every rank can still agree on the other all-reduce
        try:
            comm.probe()
        except Exception as e:
            ok, why = False, "probe all-reduce: " + str(e)
        ok = all_ok(ok)
    if not ok:
        if comm is not None:
            comm.close()
        return None, why or "another rank failed"
    return comm, ""


class ShardedLoop:
    """The update_div loop of one rank (README.md:40-54 contract; checks are summed over ranks)."""

    def __init__(self, backend, allreduce_sum: Callable, allreduce_scalars: Callable):
        self.b = backend
        self.allreduce_sum = allreduce_sum            # in-place sum over ranks of the partial buffer
        self.allreduce_scalars = allreduce_scalars    # list[float] -> list[float], summed over ranks

    def iterate(self, iters: int = 1) -> None:
        for _ in range(iters):
            self.b.update_h()
            buf = self.b.w_partial()
            self.allreduce_sum(buf)
            self.b.w_apply()

    def check(self):
        kl, d, a = self.allreduce_scalars(list(self.b.check_local()))
        return kl, (d / a if a > 0 else 0.0)

    def run(self, thresh: float = 0.0, max_iter: int = 200, iter_check: int = 25, verbose: int = 0, rank: int = 0):
        """Same convergence logic as nmf_solver_run / update_div."""
        checks = thresh > 0 or verbose
        kls = []
        prev = 0.0
        if checks:
            prev, rl1 = self.check()
            kls.append(prev)
        it = 0
        while it < max_iter:
            n = max_iter - it
            if checks:
                n = min(n, iter_check - (it % iter_check))
            self.iterate(n)
            it += n
            if checks and it % iter_check == 0:
                cur, rl1 = self.check()
                kls.append(cur)
                if verbose and rank == 0:
                    print(f"iter {it:5d}  kl-divergence {cur:.6e}  rel-L1 error {rl1:.6e}")
                stop = thresh > 0 and (prev - cur) / prev < thresh
                prev = cur
                if stop:
                    break
        return it, kls


class GpuShard:
    """Backend over the HIP solver.  The partial buffer is a torch tensor so that torch.distributed (backend
    "nccl" = RCCL over xGMI) can all-reduce it in place.  Solver and collective share one dedicated torch stream:
    torch orders a collective against the *current* stream, and the default stream's raw handle is NULL, which
    the solver would read as "create your own stream" -- so never the default stream here."""

    def __init__(self, M: int, N_local: int, K: int, device: int = 0, **solver_kw):
        import torch
        from . import api
        from .api import Solver
        if api._lib is not None and api._loaded_before_torch:
            raise RuntimeError("libnmf_mi355x.so was loaded before torch: the process now has two HIP runtimes and a torch "
                               "stream cannot be shared with it; import torch before the first nmf_gpu_amd call")
        self.torch = torch
        self.device = device
        self.stream = torch.cuda.Stream(device=device)
        assert self.stream.cuda_stream != 0
        self.solver = Solver(M, N_local, K, device=device, stream=self.stream.cuda_stream, use_graph=False, **solver_kw)
        _, cnt = self.solver.partial_buffer()
        with torch.cuda.stream(self.stream):
            self.buf = torch.zeros(cnt, dtype=torch.float32, device=f"cuda:{device}")
        self.stream.synchronize()
        self.solver.set_partial_buffer(self.buf.data_ptr(), cnt)

    def upload(self, W, H_local, X_local):
        self.solver.upload(W, H_local, X_local)

    def update_h(self):
        self.solver.update_h()

    def w_partial(self):
        self.solver.w_partial()
        return self.buf

    def w_apply(self):
        self.solver.w_apply()

    def allreduce_sum(self, t):
        """sum all-reduce of `t` over the default process group, ordered on this shard's stream"""
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            dist.all_reduce(t)

    def allreduce_scalars(self, values):
        import torch.distributed as dist
        with self.torch.cuda.stream(self.stream):
            t = self.torch.tensor(values, dtype=self.torch.float64, device=f"cuda:{self.device}")
            dist.all_reduce(t)
            return t.tolist()

    def check_local(self):
        return self.solver.check_sums()

    def download(self):
        return self.solver.download()

    def sync(self):
        self.solver.sync()

    def close(self):
        self.solver.close()
