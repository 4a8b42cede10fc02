#!/usr/bin/env python3
"""bench.py -- update_div on MI355X: iterations/s and effective GEMM TFLOP/s (8*M*N*K flop per
iteration) against the fp32 MFMA roofline, with the CPU oracle timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: BASELINE config 3, (M, N, K) = (4096, 65536, 256), fp32, W/H/X resident in HBM.
N > 1: weak scaling -- every rank owns 65536 columns of X and H (N = 4 is BASELINE config 4,
       M=4096 N=262144 R=256), W replicated, one all-reduce of [Z*H' ; rowsum(H)] per iteration.
A "step" is one full iteration (H half-step + W half-step, cuda/nmf.cu:108-109).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: 256 CU x 256 flop/clk/CU x 2.4 GHz
# HBM bytes per fused_step launch at cfg3 from rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE, separate passes;
# profiles/r01_pmc_summary.md).  bench.py cannot collect PMC itself; other shapes report null.
PMC_TRAFFIC_BYTES = {(4096, 65536, 256): {"H": 1.368e9, "W": 1.209e9}}
# SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles) of the same launches, same PMC runs
PMC_MFMA_BUSY = {(4096, 65536, 256): {"H": 0.912, "W": 0.917}}


def synth(seed, rows, cols):
    """U[0,1) fp32 from numpy's legacy MT19937 (the generator matrix_export.py:4-7 uses),
    column-major; chunked so the fp64 temporaries stay small."""
    rs = np.random.RandomState(seed)
    out = np.empty(rows * cols, dtype=np.float32)
    step = 1 << 24
    for i in range(0, out.size, step):
        out[i:i + step] = rs.rand(min(step, out.size - i))
    return out.reshape((rows, cols), order="F")


def cpu_baseline(M, Nfull, K, budget_s=20.0):
    """The oracle (`port`: our own C/OpenMP restatement; the reference has no CPU path) timed on this host's cores on
    a bounded sample: the first `Ns` columns of the same workload, through oracle_fast_update_div (the oracle's
    arithmetic arranged around its fastest SGEMM kernel, oracle/nmf_oracle_fast.c)."""
    import oracle
    Ns = min(Nfull, 8192)
    X = synth(1000, M, Ns); W = synth(0, M, K); H = synth(2000, K, Ns)
    try:
        oracle.lib(native=True)
        native = True
    except Exception:
        native = False
    cores = oracle.num_threads(native)
    t0 = time.perf_counter()
    oracle.update_div_fast(W, H, X, 1, native=native)               # warm-up (page faults, thread pool) and a time estimate
    t_one = time.perf_counter() - t0
    iters = int(max(3, min(200, budget_s / max(t_one, 1e-3))))
    t0 = time.perf_counter()
    oracle.update_div_fast(W, H, X, iters, native=native)
    t = time.perf_counter() - t0
    flops = 8.0 * M * Ns * K * iters
    tf = flops / t / 1e12
    return {"value": tf, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"oracle spec-mode update_div (oracle_fast_update_div), M={M} K={K}, first {Ns} of {Nfull} columns, {iters} iterations in {t:.1f} s"
                      f" ({'-march=native' if native else 'avx2'} build); = {tf * 1e12 / (8.0 * M * Nfull * K):.4f} full-size iterations/s",
            "iterations_per_s_full_size": tf * 1e12 / (8.0 * M * Nfull * K)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=65536, help="columns PER GPU")
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--comm", choices=["torch", "rccl"], default="torch",
                    help="N>1: all-reduce through torch.distributed (default) or in-library RCCL inside the hipGraph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for the cpu_baseline sample")
    ap.add_argument("--strong-total-N", type=int, default=0,
                    help="strong scaling instead of the default weak scaling: total column count split over the ranks "
                         "(BASELINE config 4: --strong-total-N 262144)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="rehearsal only: gloo lets several ranks share one GPU (with --same-device)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--rehearse-sharded", action="store_true",
                    help="rehearsal only: take the N>1 code path (process group, GpuShard, all-reduce) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    import torch
    import nmf_gpu_amd as ng
    ng.lib()   # fail loudly if the HIP library is missing
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    sharded = world > 1 or args.rehearse_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")

    M, Nloc, K = args.M, args.N, args.K
    if args.strong_total_N:
        shards = ng.column_shards(args.strong_total_N, world)
        Nloc = shards[rank][1] - shards[rank][0]
    Ntot = args.strong_total_N if args.strong_total_N else Nloc * world
    flops_per_iter = 8.0 * M * Ntot * K

    # synthetic inputs (data: "synthetic"), already resident in HBM before the timed region
    W = synth(0, M, K)
    X = synth(1000 + rank, M, Nloc)
    H = synth(2000 + rank, K, Nloc)

    comm = None
    shard = None
    if sharded and args.comm == "rccl":
        uid = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            uid = torch.frombuffer(bytearray(ng.Comm.unique_id()), dtype=torch.uint8).clone()
        uid = uid.cuda()
        dist.broadcast(uid, 0)
        comm = ng.Comm(bytes(uid.cpu().numpy().tobytes()), rank, world)
    if sharded and args.comm == "torch":
        # half-step protocol + torch.distributed all-reduce of the (M*K + K)-float partial buffer
        shard = ng.GpuShard(M, Nloc, K, device=local_rank)
        s = shard.solver
        loop = ng.ShardedLoop(shard, shard.allreduce_sum, shard.allreduce_scalars)
    else:
        s = ng.Solver(M, Nloc, K, use_graph=not args.no_graph, device=local_rank, comm=comm)
    s.upload(W, H, X)
    del X

    def step(n):
        if shard is None:
            s.iterate(n)
        else:
            loop.iterate(n)

    def fence():
        s.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    kl0, _ = s.check()
    step(args.warmup)
    fence()
    t0 = time.perf_counter()
    step(args.steps)
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    kl1, _ = s.check()
    if dist is not None and comm is None:
        kk = torch.tensor([kl0, kl1], dtype=torch.float64, device="cuda")
        dist.all_reduce(kk)
        kl0, kl1 = float(kk[0]), float(kk[1])

    # dominant kernel: the fused half-step (H- and W-step instantiations, 4*M*Nloc*K flop per launch each).
    # Its launch duration is measured live with hipEvent pairs around every launch, on the stream it is launched
    # on, over a second pass of the same `steps` iterations (launched eagerly: events cannot sit inside a graph
    # replay); rocprofv3 --kernel-trace of this command gives the same averages (profiles/).
    if shard is None and s.path == ng.PATH_FUSED:
        tp = s.iterate_timed(args.steps)
        ms_h, ms_w = tp["h_step"] / args.steps * 1e3, tp["w_step"] / args.steps * 1e3
        how = f"hipEvent pair around each of {args.steps} launches per kernel, eager pass of the same {args.steps} iterations after the timed region"
    else:
        reps = max(3, min(args.steps, 20))
        ms_h = s.time_piece(ng.api.T_H_STEP, reps) if s.path == ng.PATH_FUSED else float("nan")
        ms_w = s.time_piece(ng.api.T_W_STEP, reps) if s.path == ng.PATH_FUSED else float("nan")
        how = f"hipEvents around {reps} back-to-back launches on the solver stream, after the timed region"
    ms_k = max(ms_h, ms_w)
    k_flops = 4.0 * M * Nloc * K
    achieved = k_flops / (ms_k * 1e-3) / 1e12

    if rank == 0:
        its = args.steps / dt
        tflops = flops_per_iter * its / 1e12
        out = {
            "metric": "effective GEMM TFLOP/s of update_div (8*M*N*K flop per NMF iteration; iterations_per_s alongside)",
            "value": tflops, "unit": "TFLOP/s",
            "iterations_per_s": its,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong_total_N else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"update_div KL-NMF, M={M} N={Ntot} R={K} fp32"
                                   + (f" ({Nloc} columns per GPU, H/X column-sharded, W replicated, all-reduce via {args.comm})" if sharded else " (BASELINE config 3)"),
                       "M": M, "N": Ntot, "R": K, "path": "fused" if s.path == ng.PATH_FUSED else "unfused",
                       "hipgraph": (not args.no_graph) and shard is None,
                       "parallelism": f"N-sharded x{world}" if sharded else "single GPU"},
            "frac_of_fp32_mfma_peak": tflops / (PEAK_FP32_MFMA_TFLOPS * world),
            "kl_before": kl0, "kl_after": kl1,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         "traffic": PMC_TRAFFIC_BYTES.get((M, Nloc, K), {}).get("H" if ms_h >= ms_w else "W"),
                         "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/r01_pmc_summary.md); algorithmic 1.21e9",
                         "hbm_gbps": (PMC_TRAFFIC_BYTES[(M, Nloc, K)]["H" if ms_h >= ms_w else "W"] / (ms_k * 1e-3) / 1e9) if (M, Nloc, K) in PMC_TRAFFIC_BYTES else None,
                         "hbm_peak_gbps": 8000.0,
                         "mfma_busy_pmc": PMC_MFMA_BUSY.get((M, Nloc, K), {}).get("H" if ms_h >= ms_w else "W"),
                         "kernel": "%s (%s-step instantiation, the slower of the two)" % ("fused_step_kernel_v3<KT=1>" if K <= 32 else "fused_step_kernel_k16<NB=%d>" % (-(-K // 64)), "H" if ms_h >= ms_w else "W"),
                         "flop_per_launch": k_flops, "ms_per_launch": ms_k,
                         "ms_h_step": ms_h, "ms_w_step": ms_w,
                         "measured": how},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, Ntot, K, args.cpu_budget)
        print(json.dumps(out), flush=True)
    s.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
