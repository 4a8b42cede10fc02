"""N-sharded loop on CPU: world_size 2 and 3 over gloo, the oracle as the per-rank
backend, against the single-process oracle.  Exercises the column partition (ragged), the
all-reduce payload [Z*H' ; rowsum(H)], the replication of W and the rank-summed KL check."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def test_column_shards(ng):
    assert ng.column_shards(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert ng.column_shards(8, 8) == [(i, i + 1) for i in range(8)]
    assert ng.column_shards(262144, 8)[-1] == (229376, 262144)
    with pytest.raises(ValueError):
        ng.column_shards(3, 4)
    # north_star: shard only where N amortises the all-reduce
    assert ng.worth_sharding(4096, 262144, 256, 8) and ng.worth_sharding(8192, 131072, 512, 8)
    assert not ng.worth_sharding(1024, 4096, 64, 8) and not ng.worth_sharding(4096, 65536, 256, 1)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, N, K, iters, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="2")
    import torch
    import torch.distributed as dist
    import oracle
    import nmf_gpu_amd as ng
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, W, H = oracle.gen_problem(M, N, K, seed=1)
    a, b = ng.column_shards(N, world)[rank]

    class OracleShard:
        """CPU stand-in for the HIP solver's half-step protocol (test infrastructure)."""
        def __init__(self):
            self.W = oracle.clamp(W)
            self.H = oracle.clamp(np.asfortranarray(H[:, a:b]))
            self.X = oracle.clamp(np.asfortranarray(X[:, a:b]))
            self.buf = torch.zeros(M * K + K, dtype=torch.float32)

        def update_h(self):
            self.H = oracle.update_h(self.W, self.H, self.X)

        def w_partial(self):
            wh = np.maximum(oracle.sgemm("nn", self.W, self.H), oracle.EPS)
            Z = np.asfortranarray(self.X / wh)
            zht = oracle.sgemm("nt", Z, self.H)
            self.buf[: M * K] = torch.from_numpy(zht.reshape(-1, order="F").copy())
            self.buf[M * K:] = torch.from_numpy(oracle.sum_rows(self.H))
            return self.buf

        def w_apply(self):
            p = self.buf[: M * K].numpy().reshape((M, K), order="F")
            s = np.maximum(self.buf[M * K:].numpy(), oracle.EPS)
            self.W = np.asfortranarray(self.W * (p / s[None, :]))

        def check_local(self):
            wh = np.maximum(oracle.sgemm("nn", self.W, self.H), oracle.EPS)
            x, y = self.X.astype(np.float64), wh.astype(np.float64)
            return oracle.kl_div(self.X, wh), float(np.abs(x - y).sum()), float(np.abs(x).sum())

    def ar_scalars(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t)
        return t.tolist()

    sh = OracleShard()
    loop = ng.ShardedLoop(sh, lambda t: dist.all_reduce(t), ar_scalars)
    it, kls = loop.run(thresh=1e-30, max_iter=iters, iter_check=5)
    Hs = [None] * world
    dist.all_gather_object(Hs, sh.H)
    Ws = [None] * world
    dist.all_gather_object(Ws, sh.W)
    if rank == 0:
        q.put((it, kls, np.concatenate(Hs, axis=1), Ws))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,N", [(2, 96), (3, 100)])
def test_sharded_loop_matches_single_process(oracle, world, N):
    import multiprocessing as mp
    M, K, iters = 64, 8, 10
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, N, K, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    it, kls, Hcat, Ws = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, W, H = oracle.gen_problem(M, N, K, seed=1)
    Wr, Hr, itr, klr = oracle.update_div(W, H, X, 1e-30, iters, 5)
    assert it == itr == iters
    for w in Ws[1:]:
        assert np.array_equal(w, Ws[0])            # W replicated bit for bit
    assert oracle.relF(Ws[0], Wr) < 1e-5 and oracle.relF(Hcat, Hr) < 1e-5
    assert np.allclose(kls, klr, rtol=1e-6)


def _negotiate_worker(rank, world, port, fail, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    import nmf_gpu_amd as ng
    from datetime import timedelta
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=60))
    closed = []

    class FakeComm:
        def __init__(self, uid, r, w):
            if fail == ("comm", r):
                raise RuntimeError("ncclCommInitRank failed")
            self.uid, self.rank = uid, r

        def close(self):
            closed.append(self.rank)

    def unique_id():
        if fail == ("uid", 0):
            raise RuntimeError("librccl not loadable")
        return bytes(range(128))

    comm, why = ng.negotiate_comm(dist, torch, rank, world, unique_id, FakeComm, device="cpu")
    dist.barrier()                       # everybody got here: nobody is stuck in a collective the others never entered
    q.put((rank, comm is not None, comm.uid == bytes(range(128)) if comm is not None else None, why, closed))
    dist.destroy_process_group()


@pytest.mark.parametrize("fail", [None, ("uid", 0), ("comm", 1), ("comm", 0)])
def test_comm_negotiation_runs_the_same_collectives_on_every_rank_whatever_fails(fail):
    """bench.py's RCCL-or-torch decision (nmf_gpu_amd.sharded.negotiate_comm) over gloo with three ranks and a fake
    communicator: rank 0 failing to make the id, or any rank failing to build its communicator, must leave EVERY rank with
    'no communicator' (and close the ones that were built) instead of a rank waiting in a broadcast the failing rank never
    entered (round-2 advisor)."""
    import multiprocessing as mp
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_negotiate_worker, args=(r, world, port, fail, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    if fail is None:
        assert all(ok and same_uid and why == "" for _, ok, same_uid, why, _ in out)
    else:
        assert all(not ok and why for _, ok, _, why, _ in out)
        if fail[0] == "comm":            # the ranks that did build one closed it
            assert all(closed == ([r] if r != fail[1] else []) for r, _, _, _, closed in out)
