#!/usr/bin/env python3
"""bench.py -- update_div on MI355X: iterations/s and effective GEMM TFLOP/s (8*M*N*K flop per
iteration) against the fp32 MFMA roofline, with the CPU oracle timed beside it.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N = 1: BASELINE config 3, (M, N, K) = (4096, 65536, 256), fp32, W/H/X resident in HBM.
N > 1: weak scaling -- every rank owns 65536 columns of X and H (N = 4 is BASELINE config 4,
       M=4096 N=262144 R=256), W replicated, one all-reduce of [Z*H' ; rowsum(H)] per iteration
       (in-library RCCL captured inside each rank's hipGraph; --comm torch: torch.distributed, eager).
       --preset cfg4 / cfg5: the two sharded BASELINE configs as STRONG scaling (262144 resp. 131072 columns in all, split
       over the ranks; on one GPU the whole problem).  The N > 1 line also carries allreduce_ms_per_step / compute_ms_per_step
       (hipEvent pairs of an eager pass) and the RCCL library actually loaded.
A "step" is one full iteration (H half-step + W half-step, cuda/nmf.cu:108-109).
Protocol: W warm-up steps, then `--repeats` (5) timed regions of exactly K steps each, every region
bracketed by barrier + synchronize and maxed over ranks; `value` / `ms_per_step` are the MEDIAN region, all regions are
listed in `repeats_ms_per_step`.  Inputs: X -> W -> H from one MT19937 stream, seed 0 (rank r > 0: X_r, H_r from seed r,
W from seed 0's stream position), column-major, as matrix_export.py:4-7 extended to other shapes.
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
# HBM traffic is a PMC quantity (rocprofv3 --pmc passes of their own, collected and corrected as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE doubled on gfx950): this script does not collect counters.  `roofline.traffic` is the bytes per launch of the dominant
# kernel from the committed PMC passes of this very command (profiles/pmc_static.json, written by tools/pmc_summary.py --json),
# reported with its source; dividing it by THIS run's ms_per_launch gives `hbm_tbps` next to `algorithmic_tbps`.
def pmc_static(M, N, K):
    """the committed PMC passes of this shape, or None; with "stale": <why> when they were taken of other kernel sources than this
    tree's (tools/kernel_identity.py) -- `traffic` is then reported as null instead of next to live timings it does not belong to"""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_static.json")) as f:
            e = json.load(f).get(f"{M}x{N}x{K}")
        if e is None:
            return None
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from kernel_identity import kernel_sources_sha16
        have, want = e.get("kernel_sources_sha16"), kernel_sources_sha16()
        if have != want:
            e = dict(e, stale=f"the committed PMC passes are of kernel sources {have or 'of an unrecorded build'}, this tree's are {want}: traffic not quoted")
        return e
    except Exception:
        return None


# X once + the owned factor in and out + the streamed factor once (it is re-read from L2): algorithmic bytes of one half-step
def algorithmic_bytes(M, N, K):
    return 4.0 * (M * N + 2.0 * K * N + M * K), 4.0 * (M * N + 2.0 * M * K + K * N)   # H-step, W-step


def _draw(rs, rows, cols, keep=True):
    """rows*cols values of U[0,1) fp32 from the legacy MT19937 stream `rs` (what matrix_export.py:4-7 uses), the flat
    buffer read column-major; chunked so the fp64 temporaries stay small.  keep=False only advances the stream."""
    n = rows * cols
    out = np.empty(n, dtype=np.float32) if keep else None
    step = 1 << 24
    for i in range(0, n, step):
        v = rs.rand(min(step, n - i))
        if keep:
            out[i:i + step] = v
    return out.reshape((rows, cols), order="F") if keep else None


def synth_problem(rank, M, Nloc, K):
    """X -> W -> H from ONE stream.  Rank 0 / single GPU: seed 0, exactly matrix_export.py's recipe at this shape.
    Rank r > 0 draws its own column block X_r, H_r from seed r and takes the replicated W from seed 0's stream position."""
    rs = np.random.RandomState(rank)
    X = _draw(rs, M, Nloc)
    W = _draw(rs, M, K)
    H = _draw(rs, K, Nloc)
    if rank != 0:
        r0 = np.random.RandomState(0)
        _draw(r0, M, Nloc, keep=False)
        W = _draw(r0, M, K)
    return X, W, H


def cpu_baseline(M, Nfull, K, budget_s=20.0, full=None):
    """The oracle (`port`: our own C/OpenMP restatement; the reference has no CPU path) timed on this host's cores through
    oracle_fast_update_div (the oracle's arithmetic arranged around its fastest SGEMM kernel, oracle/nmf_oracle_fast.c).
    `full` = the (X, W, H) of the benchmark itself: the headline figure is then >= 3 iterations of the WHOLE problem, all Nfull
    columns (SURVEY 8d / BASELINE.md 3: "at cfg3+ time >= 3 iterations"), and the 8192-column sample of earlier rounds rides along
    as `sample_8192`; without `full` (N > 1 ranks never call this) only the sample is timed."""
    import oracle
    try:
        oracle.lib(native=True)
        native = True
    except Exception:
        native = False
    cores = oracle.num_threads(native)
    build = "-march=native" if native else "avx2"

    def timed(X, W, H, budget, lo, hi):
        t0 = time.perf_counter()
        oracle.update_div_fast(W, H, X, 1, native=native)               # warm-up (page faults, thread pool) and a time estimate
        t_one = time.perf_counter() - t0
        iters = int(max(lo, min(hi, budget / max(t_one, 1e-3))))
        t0 = time.perf_counter()
        oracle.update_div_fast(W, H, X, iters, native=native)
        t = time.perf_counter() - t0
        return 8.0 * M * X.shape[1] * K * iters / t / 1e12, iters, t

    what = ("oracle spec-mode update_div through oracle_fast_update_div (the oracle's arithmetic arranged around its fastest SGEMM kernel; equal to the "
            "golden-pinned loop to 5e-6 relF, tests/test_oracle_ops.py, not the pinned routine itself)")
    Ns = min(Nfull, 8192)
    if full is not None and full[0].shape[1] == Nfull and Nfull > Ns:
        Xs, Ws, Hs = full[0][:, :Ns], full[1], full[2][:, :Ns]          # the same generator, seed 0: the benchmark's own first columns
        tf_s, it_s, t_s = timed(np.asfortranarray(Xs), Ws, np.asfortranarray(Hs), 0.25 * budget_s, 3, 200)
        tf, iters, t = timed(full[0], full[1], full[2], 0.5 * budget_s, 3, 12)
        return {"value": tf, "unit": "TFLOP/s", "cores": cores, "kind": "port",
                "sample": f"{what}, M={M} K={K}, {Nfull} of {Nfull} columns (the benchmark's own X, W, H), {iters} iterations in {t:.1f} s after one warm-up iteration"
                          f" ({build} build); = {tf * 1e12 / (8.0 * M * Nfull * K):.4f} full-size iterations/s",
                "iterations": iters, "iterations_per_s_full_size": tf * 1e12 / (8.0 * M * Nfull * K),
                "sample_8192": {"value": tf_s, "unit": "TFLOP/s", "sample": f"{Ns} of {Nfull} columns, {it_s} iterations in {t_s:.1f} s"}}
    X, W, H = synth_problem(0, M, Ns, K) if full is None else (np.asfortranarray(full[0][:, :Ns]), full[1], np.asfortranarray(full[2][:, :Ns]))
    tf, iters, t = timed(X, W, H, budget_s, 3, 200)
    return {"value": tf, "unit": "TFLOP/s", "cores": cores, "kind": "port",
            "sample": f"{what}, M={M} K={K}, {Ns} of {Nfull} columns (same generator, seed 0), {iters} iterations in {t:.1f} s"
                      f" ({build} build); = {tf * 1e12 / (8.0 * M * Nfull * K):.4f} full-size iterations/s",
            "iterations": iters, "iterations_per_s_full_size": tf * 1e12 / (8.0 * M * Nfull * K)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--M", type=int, default=4096)
    ap.add_argument("--N", type=int, default=65536, help="columns PER GPU")
    ap.add_argument("--K", type=int, default=256)
    ap.add_argument("--repeats", type=int, default=5, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--preset", choices=["cfg3", "cfg2", "gold", "gold100", "gold200", "paper", "cfg4", "cfg5"], default="cfg3",
                    help="cfg3 (default, BASELINE config 3: the metric's configuration); cfg2 = 1024 x 4096 x 64 (BASELINE config 2), gold = the "
                         "reference's own 4096 x 350 x 128 (matrix_export.py:4-7; gold100 / gold200: the same problem at R = 100 / 200, ranks between the "
                         "powers of two), paper = 512 x 3445 x 30: the same JSON line for the small shapes; "
                         "cfg4 = 4096 x 262144 x 256 and cfg5 = 8192 x 131072 x 512 (BASELINE configs 4, 5): strong scaling, the columns split over --gpus")
    ap.add_argument("--comm", choices=["auto", "torch", "rccl"], default="auto",
                    help="N>1: all-reduce by in-library RCCL captured inside the per-iteration hipGraph (rccl; auto = rccl, falling "
                         "back to torch if the communicator cannot be set up) or through torch.distributed, eager (torch)")
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

    if args.preset in ("cfg4", "cfg5"):
        args.M, args.strong_total_N, args.K = {"cfg4": (4096, 262144, 256), "cfg5": (8192, 131072, 512)}[args.preset]
        args.N = args.strong_total_N
    elif args.preset != "cfg3":
        args.M, args.N, args.K = {"cfg2": (1024, 4096, 64), "gold": (4096, 350, 128), "gold100": (4096, 350, 100), "gold200": (4096, 350, 200),
                                  "paper": (512, 3445, 30)}[args.preset]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world

    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this pool
    # The contract is ONE JSON line on stdout.  RCCL prints a version banner on stdout when its first communicator comes up
    # (torch's bundled 2.26.6 does, on every rank): keep the real stdout aside for the JSON line and send everything else that
    # any library writes to file descriptor 1 to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
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
    X, W, H = synth_problem(rank, M, Nloc, K)

    comm = None
    shard = None
    comm_used = args.comm
    if sharded and args.comm in ("auto", "rccl"):
        # every rank runs the same torch collectives on the success and the failure path (nmf_gpu_amd.sharded.negotiate_comm)
        comm, why = ng.negotiate_comm(dist, torch, rank, world, ng.Comm.unique_id, ng.Comm, device="cuda")
        if comm is None:
            if args.comm == "rccl":
                raise SystemExit(f"bench.py rank {rank}: --comm rccl: the in-library RCCL communicator could not be set up on every rank ({why})")
            print(f"bench.py rank {rank}: in-library RCCL unavailable ({why}); using torch.distributed", file=sys.stderr)
            comm_used = "torch"
        else:
            comm_used = "rccl"
    if sharded and comm_used == "torch":
        # half-step protocol + torch.distributed all-reduce of the (M*K + K)-float partial buffer
        shard = ng.GpuShard(M, Nloc, K, device=local_rank)
        s = shard.solver
        loop = ng.ShardedLoop(shard, shard.allreduce_sum, shard.allreduce_scalars)
    else:
        s = ng.Solver(M, Nloc, K, use_graph=not args.no_graph, device=local_rank, comm=comm)
    s.upload(W, H, X)
    if comm is not None:
        # The first iterations through the in-library communicator -- graph capture with the all-reduce inside, replay, an all-reduced
        # check (which waits with the communicator's deadline, NMF_COMM_TIMEOUT_S) -- while every rank can still agree on the other path:
        # N > 1 ranks over xGMI have never run before the driver's scaling run, and a bench line from torch.distributed beats none.
        ok, why = True, ""
        try:
            s.prepare(args.steps)
            s.iterate(max(1, args.warmup))
            s.check()
        except Exception as e:
            ok, why = False, str(e)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) != 1:
            print(f"bench.py rank {rank}: the first iterations over the in-library communicator failed ({why or 'on another rank'}); using torch.distributed", file=sys.stderr)
            if args.comm == "rccl":
                raise SystemExit(f"bench.py rank {rank}: --comm rccl: the in-library RCCL all-reduce does not work here ({why or 'another rank failed'})")
            try:
                s.close(); comm.close()
            except Exception:
                pass
            comm, comm_used = None, "torch"
            shard = ng.GpuShard(M, Nloc, K, device=local_rank)
            s = shard.solver
            loop = ng.ShardedLoop(shard, shard.allreduce_sum, shard.allreduce_scalars)
        s.upload(W, H, X)      # the timed run starts from the same factors either way
    host_problem = (X, W, H) if (world == 1 and not args.no_cpu_baseline) else None   # cpu_baseline times the benchmark's own problem
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
    if shard is None:
        s.prepare(args.steps)      # hipGraph capture + instantiation of what the timed regions replay: not part of any step
    step(args.warmup)
    fence()
    region = []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        step(args.steps)
        fence()
        dt_r = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt_r], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_r = float(tt.item())
        region.append(dt_r)
    dt = float(np.median(region))
    kl1, _ = s.check()
    if dist is not None and comm is None:
        kk = torch.tensor([kl0, kl1], dtype=torch.float64, device="cuda")
        dist.all_reduce(kk)
        kl0, kl1 = float(kk[0]), float(kk[1])

    # dominant kernel: the fused half-step (H- and W-step instantiations, 4*M*Nloc*K flop per launch each).
    # An eager pass of the same `steps` iterations after the timed regions brackets every launch with a hipEvent pair on the
    # stream it is launched on (events cannot sit inside a graph replay).  The event records between launches cost a few
    # per cent, so the eager figures are reported as measured (`eager_event_*`) and the roofline's ms_per_launch is the
    # kernel's SHARE of the eager pass applied to the timed region's ms_per_step: the per-launch figures then add up to at
    # most the step the bench line reports.  rocprofv3 --kernel-trace of this command gives the same averages (profiles/).
    ms_step = dt / args.steps * 1e3
    pieces = None
    if shard is None and s.path == ng.PATH_FUSED:
        tp = s.iterate_timed(args.steps)
        pieces = {k: tp[k] / args.steps * 1e3 for k in ("h_step", "w_step", "sums", "apply", "allreduce")}
        eager_h, eager_w = pieces["h_step"], pieces["w_step"]
        how = (f"hipEvent pair around each of {args.steps} launches per kernel in an eager pass of the same {args.steps} iterations after the timed "
               f"regions; ms_per_launch = that kernel's share of the eager pass x the timed region's ms_per_step")
    else:
        reps = max(3, min(args.steps, 20))
        eager_h = s.time_piece(ng.api.T_H_STEP, reps) if s.path == ng.PATH_FUSED else float("nan")
        eager_w = s.time_piece(ng.api.T_W_STEP, reps) if s.path == ng.PATH_FUSED else float("nan")
        how = f"hipEvents around {reps} back-to-back launches on the solver stream, after the timed region"
    if pieces is not None and sum(pieces.values()) > 0:
        scale = min(1.0, ms_step / sum(pieces.values()))
        ms_h, ms_w = eager_h * scale, eager_w * scale
    else:
        ms_h, ms_w = eager_h, eager_w
    ms_k = max(ms_h, ms_w)
    k_flops = 4.0 * M * Nloc * K
    achieved = k_flops / (ms_k * 1e-3) / 1e12
    which = "H" if ms_h >= ms_w else "W"
    pmc = pmc_static(M, Nloc, K)
    alg_bytes = algorithmic_bytes(M, Nloc, K)[0 if which == "H" else 1]
    traffic = pmc["hbm_bytes_per_launch"][which] if (pmc and not pmc.get("stale")) else None
    if shard is not None:
        # torch.distributed path: event pairs on the shard's stream around the all-reduce and around the whole iteration
        ev = []
        with torch.cuda.stream(shard.stream):
            for _ in range(args.steps):
                e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                e[0].record(shard.stream); shard.update_h(); buf = shard.w_partial()
                e[1].record(shard.stream); shard.allreduce_sum(buf)
                e[2].record(shard.stream); shard.w_apply()
                e[3].record(shard.stream)
                ev.append(e)
        shard.stream.synchronize()
        ar_t = sum(e[1].elapsed_time(e[2]) for e in ev) / args.steps
        tot_t = sum(e[0].elapsed_time(e[3]) for e in ev) / args.steps
        pv = torch.tensor([ar_t, tot_t - ar_t], dtype=torch.float64, device="cuda")
        dist.all_reduce(pv, op=dist.ReduceOp.MAX)
        ar_ms, comp_ms = float(pv[0]), float(pv[1])
    elif dist is not None and pieces is not None:      # the slowest rank's figures
        pv = torch.tensor([pieces["allreduce"], pieces["h_step"] + pieces["w_step"] + pieces["sums"] + pieces["apply"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(pv, op=dist.ReduceOp.MAX)
        ar_ms, comp_ms = float(pv[0]), float(pv[1])
    elif pieces is not None:
        ar_ms, comp_ms = pieces["allreduce"], pieces["h_step"] + pieces["w_step"] + pieces["sums"] + pieces["apply"]
    else:
        ar_ms, comp_ms = None, None

    if rank == 0:
        its = args.steps / dt
        tflops = flops_per_iter * its / 1e12
        out = {
            "metric": "effective GEMM TFLOP/s of update_div (8*M*N*K flop per NMF iteration; iterations_per_s alongside)",
            "value": tflops, "unit": "TFLOP/s",
            "iterations_per_s": its,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "timed_repeats": len(region), "repeats_ms_per_step": [r / args.steps * 1e3 for r in region],
            "ms_per_step_min": min(region) / args.steps * 1e3, "ms_per_step_max": max(region) / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.strong_total_N else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"update_div KL-NMF, M={M} N={Ntot} R={K} fp32"
                                   + (f" ({Nloc} columns per GPU, H/X column-sharded, W replicated, all-reduce via {comm_used})" if sharded
                                      else {"cfg3": " (BASELINE config 3)", "cfg2": " (BASELINE config 2)", "gold": " (the reference's own problem)", "paper": " (the paper's example)",
                                            "gold100": " (the reference's own problem at R = 100)", "gold200": " (the reference's own problem at R = 200)",
                                            "cfg4": " (BASELINE config 4, whole on one GPU)", "cfg5": " (BASELINE config 5, whole on one GPU)"}[args.preset]
                                      if (M, Nloc, K) in ((4096, 65536, 256), (1024, 4096, 64), (4096, 350, 128), (4096, 350, 100), (4096, 350, 200), (512, 3445, 30),
                                                          (4096, 262144, 256), (8192, 131072, 512)) else ""),
                       "M": M, "N": Ntot, "R": K, "path": "fused" if s.path == ng.PATH_FUSED else "unfused",
                       "hipgraph": (not args.no_graph) and shard is None,
                       "parallelism": f"N-sharded x{world}" if sharded else "single GPU"},
            "frac_of_fp32_mfma_peak": tflops / (PEAK_FP32_MFMA_TFLOPS * world),
            "kl_before": kl0, "kl_after": kl1,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                         # the fraction of peak of the WHOLE iteration (8*M*N*K flop over ms_per_step: helper launches, launch gaps and
                         # the shorter half-step included) -- the figure to quote for a shape; `frac` is the dominant launch alone
                         "frac_whole_iteration": tflops / (PEAK_FP32_MFMA_TFLOPS * world),
                         # HBM bytes per launch of the dominant kernel: PMC counters of the committed passes of this command
                         # (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md), not collected by this run; null where no pass is committed
                         "traffic": traffic,
                         "traffic_source": (pmc.get("stale") or pmc["source"]) if pmc else None,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                         "hbm_tbps": (traffic / (ms_k * 1e-3) / 1e12) if traffic else None,
                         "algorithmic_tbps": alg_bytes / (ms_k * 1e-3) / 1e12,
                         "mfma_busy_frac_of_simd_cycles": pmc["mfma_busy_frac_of_simd_cycles"][which] if (pmc and not pmc.get("stale")) else None,
                         "kernel": "%s (%s-step launch, the slower of the two)" % (s.describe(), which),
                         "flop_per_launch": k_flops, "ms_per_launch": ms_k,
                         "ms_h_step": ms_h, "ms_w_step": ms_w,
                         "eager_event_ms_h_step": eager_h, "eager_event_ms_w_step": eager_w,
                         "eager_event_ms_per_piece": pieces,
                         "measured": how},
        }
        if sharded:
            # where an N > 1 step goes: the slowest rank's all-reduce and compute time per step in the eager, event-bracketed
            # pass (in the graph-replayed timed regions the two overlap nothing either: the all-reduce sits between the W-step's
            # partial product and its apply), and which RCCL the library actually loaded
            out["allreduce_ms_per_step"] = ar_ms
            out["compute_ms_per_step"] = comp_ms
            out["allreduce_bytes"] = 4 * (M * K + K)
            out["comm"] = comm_used
            try:
                out["rccl"] = ng.comm_library_info() if comm_used == "rccl" else f"torch.distributed ({args.dist_backend})"
            except Exception as e:
                out["rccl"] = f"unavailable ({e})"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(M, Ntot, K, args.cpu_budget, full=host_problem)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    s.close()
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
