"""N-sharded loop on the GPU path: two ranks share the one GPU of the test box (gloo moves the partial buffer),
each driving its own HIP solver through GpuShard/ShardedLoop exactly as bench.py does for N > 1.  Checks the
stream ordering between the solver's kernels and the collective, the replication of W and parity with the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, M, N, K, iters, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
    import torch
    import torch.distributed as dist
    import oracle
    import nmf_gpu_amd as ng
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    X, W, H = oracle.gen_problem(M, N, K, seed=1)
    a, b = ng.column_shards(N, world)[rank]
    shard = ng.GpuShard(M, b - a, K, device=0)
    shard.upload(W, np.asfortranarray(H[:, a:b]), np.asfortranarray(X[:, a:b]))
    loop = ng.ShardedLoop(shard, shard.allreduce_sum, shard.allreduce_scalars)
    it, kls = loop.run(thresh=1e-30, max_iter=iters, iter_check=5)
    Wl, Hl = shard.download()
    Hs, Ws = [None] * world, [None] * world
    dist.all_gather_object(Hs, Hl)
    dist.all_gather_object(Ws, Wl)
    if rank == 0:
        q.put((it, kls, np.concatenate(Hs, axis=1), Ws))
    dist.barrier()
    shard.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("N", [512, 333])
def test_two_ranks_one_gpu_match_oracle(oracle, N):
    import multiprocessing as mp   # not torch.multiprocessing: torch must not enter this process after libnmf (two HIP runtimes)
    M, K, iters, world = 256, 64, 10, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, M, N, K, iters, q)) for r in range(world)]
    for p in procs:
        p.start()
    it, kls, Hcat, Ws = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    X, W, H = oracle.gen_problem(M, N, K, seed=1)
    Wr, Hr, itr, klr = oracle.update_div(W, H, X, 1e-30, iters, 5)
    assert it == itr == iters
    assert np.array_equal(Ws[0], Ws[1])                       # W stays replicated bit for bit
    assert oracle.relF(Ws[0], Wr) < 1e-5 and oracle.relF(Hcat, Hr) < 1e-5
    assert np.allclose(kls, klr, rtol=1e-5)
