"""The multi-device paths of the library on the one GPU of a test box: (1) the N-sharded update_div driver (nmf_multi.cpp) with
emulated shards -- every rank the same collective sequence, deadlines on every wait, group abort, the automatic fallback to
one GPU; (2) "replicas only" multi-restart NMF over several workers (update_div_restarts with a device list), SURVEY 8e / 8f4,
paper section 3.2.  Faults are injected through environment variables the library reads per call (NMF_FAULT_*)."""
import os
import subprocess
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _rand_f(rng, rows, cols):
    """uniform [0, 1) fp32, column-major, without the transposing copy np.asfortranarray(rng.random((rows, cols))) makes (seconds per GiB)"""
    return rng.random((cols, rows), dtype=np.float32).T
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _env:
    def __init__(self, **kw):
        self.kw, self.old = kw, {}

    def __enter__(self):
        for k, v in self.kw.items():
            self.old[k] = os.environ.get(k)
            os.environ[k] = str(v)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("M,N,K,G", [(1024, 4096, 64, 2), (300, 2000, 128, 3)])
def test_sharded_run_with_verbose_and_no_threshold_issues_the_same_collectives_on_every_rank(ng, oracle, capfd, M, N, K, G):
    """verbose = 1 with CONVERGE_THRESH = 0 turns the KL checks on (cuda/nmf.cu:9-11 + README.md:54).  Each check all-reduces
    three doubles, so EVERY rank has to evaluate them although only rank 0 prints -- ranks that skipped them paired their
    next f32 all-reduce with rank 0's f64 one (round-2 advisor finding: corrupt W or a hang).  The drop-in call
    update_div(W, H, X, 0, 200, NULL, 1) takes exactly this path once it shards."""
    X, W, H = oracle.gen_problem(M, N, K, seed=31)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    with _env(NMF_COMM_TIMEOUT_S=20):
        r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=60, verbose=1, emulate_shards=G, iter_check=25)
    out = capfd.readouterr().out
    assert r["n_shards"] == G and r["w_replicas_identical"] == 1 and r["iterations"] == 60 and len(r["kl"]) == 3
    assert out.count("kl-divergence") == 3                     # iteration 0, 25, 50: printed once, by rank 0
    Wr, Hr, _, klr = oracle.update_div(W, H, X, 1e-30, 60, 25)
    assert oracle.relF(Wm.mat, Wr) < 1e-5 and oracle.relF(Hm.mat, Hr) < 1e-5
    assert np.allclose(r["kl"], klr[:3], rtol=2e-5)


def test_a_rank_that_never_reaches_its_first_collective_aborts_the_group_within_the_deadline(ng, oracle):
    """VERDICT r02 'deadline on the first collective': rank 1 of 2 stalls before its first all-reduce (NMF_FAULT_STALL_RANK).
    Rank 0's wait runs into NMF_COMM_TIMEOUT_S, aborts the group, every rank returns, the call reports NMF_ERR_COMM -- and
    the caller's W.mat / H.mat are untouched (they are written only by a run that succeeded on every rank)."""
    M, N, K = 256, 1024, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=32)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    t0 = time.time()
    with _env(NMF_FAULT_STALL_RANK=1, NMF_COMM_TIMEOUT_S=2):
        with pytest.raises(ng.NmfError) as e:
            ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=20, emulate_shards=2)
    assert e.value.status == 6 and time.time() - t0 < 30
    assert np.array_equal(Wm.mat, W) and np.array_equal(Hm.mat, H)


def test_a_rank_blocked_inside_a_host_call_is_freed_by_the_watchdog(ng, oracle, capfd):
    """RCCL connects its transports inside the first collective's ENQUEUE and waits there for its peers: a rank blocked like
    that never reaches the deadline of its own wait.  NMF_FAULT_BLOCKING_COLLECTIVE takes the deadline off the emulated
    rendezvous (rank 0 blocks in it as in such a host call) while rank 1 never arrives: only the calling thread's watchdog
    (three time-outs without a heartbeat from a rank in the loop) can abort the group.  NMF_ERR_COMM, factors untouched."""
    M, N, K = 256, 1024, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=38)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    t0 = time.time()
    with _env(NMF_FAULT_BLOCKING_COLLECTIVE=1, NMF_FAULT_STALL_RANK=1, NMF_COMM_TIMEOUT_S=1):
        with pytest.raises(ng.NmfError) as e:
            ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=20, emulate_shards=2)
    dt = time.time() - t0
    assert e.value.status == 6 and 2.5 < dt < 30, dt
    assert "has not moved" in capfd.readouterr().err
    assert np.array_equal(Wm.mat, W) and np.array_equal(Hm.mat, H)


def test_automatic_sharding_falls_back_to_one_gpu_in_the_same_process(ng, oracle, capfd):
    """the same stall when sharding was the library's own idea (n_devices = 0; NMF_EMULATE_SHARDS stands in for the devices of
    a multi-GPU node): the drop-in call must not hang and must not fail -- it reports the aborted sharded run on stderr and
    runs the problem on one GPU, from the caller's untouched factors: the result equals a plain one-GPU call bit for bit"""
    M, N, K = 512, 2048, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=33)
    W1, H1 = ng.Matrix(W), ng.Matrix(H)
    r1 = ng.update_div_ex(W1, H1, ng.Matrix(X), max_iter=30, n_devices=1)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    with _env(NMF_EMULATE_SHARDS=2, NMF_FAULT_STALL_RANK=0, NMF_COMM_TIMEOUT_S=2):
        r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=30)
    err = capfd.readouterr().err
    assert "running on one GPU" in err
    assert r["n_shards"] == 1 and r["iterations"] == r1["iterations"] == 30
    assert np.array_equal(Wm.mat, W1.mat) and np.array_equal(Hm.mat, H1.mat)
    # without the fault the environment-selected shards run, and agree with the one-GPU result to the all-reduce's reordering
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    with _env(NMF_EMULATE_SHARDS=2):
        r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=30)
    assert r["n_shards"] == 2 and oracle.relF(Wm.mat, W1.mat) < 1e-5 and oracle.relF(Hm.mat, H1.mat) < 1e-5


def test_an_all_reduce_that_fails_mid_run_on_one_rank_ends_every_rank(ng, oracle):
    """the 7th all-reduce of rank 1 fails (NMF_FAULT_ALLREDUCE=1:7: warm-up + six iterations in): that rank aborts the group,
    rank 0 -- waiting inside the matching collective -- is woken, the call returns NMF_ERR_COMM instead of hanging, and nothing
    half-iterated reaches the caller's matrices (round-2 advisor: ERR_COMM after a partial gather must not be resumed from)"""
    M, N, K = 256, 1024, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=34)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    t0 = time.time()
    with _env(NMF_FAULT_ALLREDUCE="1:7", NMF_COMM_TIMEOUT_S=5):
        with pytest.raises(ng.NmfError) as e:
            ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=40, emulate_shards=2)
    assert e.value.status == 6 and time.time() - t0 < 30
    assert np.array_equal(Wm.mat, W) and np.array_equal(Hm.mat, H)
    # the group is gone with the call: the next sharded run on the same thread is unaffected
    r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=5, emulate_shards=2)
    assert r["n_shards"] == 2 and r["iterations"] == 5


def test_nan_in_one_shard_with_padding_rows_and_both_splits(ng, oracle):
    """Sharded split path with M not a multiple of 128 (rows Mv..Mp of W are zero padding, written by no workgroup) and both
    half-steps split over workgroups, so the W-step's slabs share their buffer with the H-step's (round-2 advisor: the sum
    behind the all-reduce read those never-written padding rows, i.e. whatever the H-step's slabs had left there; it now
    writes zeros without reading them).  The shape had no test.  A NaN in X poisons one column of H, then all of W
    (cuda/matrix.cu:185-186 lets NaN through the clamp) -- exactly what the oracle says; a clean problem matches the oracle."""
    M, N, K = 200, 1200, 64          # Mp = 256 > Mv = 224; per-shard N = 600: ns_h > 1 and ns_w > 1
    X, W, H = oracle.gen_problem(M, N, K, seed=35)
    X = X.copy(order="F")
    X[5, 7] = np.nan
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=1, emulate_shards=2, nsplit_h=2, nsplit_w=2, split_kernel=1)
    assert r["n_shards"] == 2
    Hr = oracle.update_h(oracle.clamp(W), oracle.clamp(H), oracle.clamp(X))
    assert np.array_equal(np.isnan(Hm.mat), np.isnan(Hr)) and np.isnan(Hr[:, 7]).all() and not np.isnan(np.delete(Hr, 7, axis=1)).any()
    ok = ~np.isnan(Hr)
    assert oracle.relF(Hm.mat[ok], Hr[ok]) < 1e-5
    assert np.isnan(Wm.mat).all()                       # every row of Z*H' sees column 7
    # a clean problem of the same shape: finite, equal to the oracle, through three iterations (stale slabs would show by then)
    X2, W2, H2 = oracle.gen_problem(M, N, K, seed=36)
    Wm, Hm = ng.Matrix(W2), ng.Matrix(H2)
    ng.update_div_ex(Wm, Hm, ng.Matrix(X2), max_iter=3, emulate_shards=2, nsplit_h=2, nsplit_w=2, split_kernel=1)
    Wr, Hr, _, _ = oracle.update_div(W2, H2, X2, 0.0, 3, 25)
    assert oracle.relF(Wm.mat, Wr) < 1e-5 and oracle.relF(Hm.mat, Hr) < 1e-5


def test_one_named_device_is_that_device_under_the_automatic_choice(ng, oracle):
    """update_div_ex(device = d) / `nmf --device d` on a multi-GPU host used to become 'shard from ordinal d', an argument
    error for d > 0 (round-2 advisor).  A caller that names one device gets that device.  On a one-GPU box: device = 0 with
    NMF_DEVICES unset and a shape worth sharding must stay a one-shard run and succeed."""
    M, N, K = 256, 512, 32
    X, W, H = oracle.gen_problem(M, N, K, seed=37)
    old = os.environ.pop("NMF_DEVICES", None)
    try:
        r = ng.update_div_ex(ng.Matrix(W), ng.Matrix(H), ng.Matrix(X), max_iter=3, device=0)
        assert r["n_shards"] == 1
    finally:
        if old is not None:
            os.environ["NMF_DEVICES"] = old


def test_rccl_library_identity_is_reported_and_checked(ng):
    """the RCCL actually dlopen()ed (under PyTorch: torch's bundled build, not /opt/rocm's) must be named, and its major
    version must match the rccl.h the function table was typed from"""
    info = ng.comm_library_info()
    assert info.startswith("RCCL 2.") and "librccl" in info and "built against rccl.h 2." in info, info


# ------------------------------------------------------------------------------ replicas-only restarts
def _pairs(M, N, K, R, seed):
    rng = np.random.default_rng(seed)
    Ws = [_rand_f(rng, M, K) for _ in range(R)]
    Hs = [_rand_f(rng, K, N) for _ in range(R)]
    return Ws, Hs


@pytest.mark.parametrize("M,N,K,R,devices,thresh", [(512, 1000, 30, 7, [0, 0], 0.0), (1024, 350, 128, 5, [0, 0, 0], 2e-3),
                                                     (256, 2048, 320, 3, [0, 0], 0.0), (2048, 2048, 256, 3, [0, 0], 0.0)])
def test_restarts_dealt_to_several_workers_equal_the_one_device_call_bit_for_bit(ng, oracle, M, N, K, R, devices, thresh):
    """SURVEY 8f4 / 8e 'replicas only': update_div_restarts(n_devices = G, devices = [...]) deals restart i to worker i % G --
    a host thread, a batched solver, a copy of X each; no communicator.  Two (three) workers on the one GPU of the box are
    what can be run of it here.  Every restart must come out exactly as in the one-device call (same kernels, same split
    counts: the split is chosen from the restart count of the whole call), with the same KL values and the same winner.
    K = 320 and 2048 x 2048 x 256 run batched on the 64-column kernel; in the latter (K <= M / 8) the W-step's normaliser comes off
    the stream (vsum_part), and with three restarts over two workers one worker is dealt a SINGLE restart: it must take the batched
    route too (round-4 ADVICE: it used to fall to the lone-problem route, whose row sums are summed in another order)."""
    X, _, _ = oracle.gen_problem(M, N, K, seed=41)
    Ws, Hs = _pairs(M, N, K, R, 42)
    W1, H1 = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
    b1, kl1 = ng.update_div_restarts(W1, H1, ng.Matrix(X), max_iter=50, converge_thresh=thresh, iter_check=10, n_devices=1)
    Wg, Hg = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
    bg, klg = ng.update_div_restarts(Wg, Hg, ng.Matrix(X), max_iter=50, converge_thresh=thresh, iter_check=10,
                                     n_devices=len(devices), devices=devices)
    assert bg == b1 and klg == kl1
    for i in range(R):
        assert np.array_equal(Wg[i].mat, W1[i].mat) and np.array_equal(Hg[i].mat, H1[i].mat), i
    # one pair against the oracle (the batched split differs from a lone update_div's: parity is with the oracle, not with its bits)
    wr, hr, _, _ = oracle.update_div(Ws[2], Hs[2], X, thresh, 50, 10)
    assert oracle.relF(Wg[2].mat, wr) < 2e-5 and oracle.relF(Hg[2].mat, hr) < 2e-5


def test_more_restarts_than_one_batch_holds(ng, oracle):
    """70 restarts: a batched solver carries at most 64 pairs, so the call makes a second pass of six with the other pairs frozen
    (set_active); every restart against the oracle, the winner the arg-min, also when the restarts are dealt to two workers"""
    M, N, K, R = 96, 160, 16, 70
    X, _, _ = oracle.gen_problem(M, N, K, seed=47)
    Ws, Hs = _pairs(M, N, K, R, 48)
    W1, H1 = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
    b1, kl1 = ng.update_div_restarts(W1, H1, ng.Matrix(X), max_iter=30, n_devices=1)
    worst = 0.0
    for i in range(R):
        wr, hr, _, _ = oracle.update_div(Ws[i], Hs[i], X, 0.0, 30, 25)
        worst = max(worst, oracle.relF(W1[i].mat, wr), oracle.relF(H1[i].mat, hr))
    assert worst < 2e-5 and b1 == int(np.argmin(kl1))
    W2, H2 = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
    b2, kl2 = ng.update_div_restarts(W2, H2, ng.Matrix(X), max_iter=30, n_devices=2, devices=[0, 0])
    assert b2 == b1 and kl2 == kl1 and all(np.array_equal(a.mat, b.mat) for a, b in zip(W1 + H1, W2 + H2))


def test_a_batch_that_does_not_fit_the_device_is_run_in_more_passes(ng, oracle):
    """round-4 ADVICE: a batched solver holds B copies of W, H and the slabs; where the device cannot hold them the call used to fail.
    With the arena capped (NMF_FAULT_ARENA_LIMIT_MB: 12 pairs of 512 x 1024 x 64 do not fit, 6 do) the call halves the pairs per
    pass; the cuts follow the restart count of the whole call, so every restart comes out bit for bit as in the unrestricted call."""
    M, N, K, R = 512, 1024, 64, 12
    X, _, _ = oracle.gen_problem(M, N, K, seed=51)
    Ws, Hs = _pairs(M, N, K, R, 52)
    W1, H1 = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
    b1, kl1 = ng.update_div_restarts(W1, H1, ng.Matrix(X), max_iter=40, n_devices=1)
    pair_mb = 4.0 * (M * K + K * N) / 1048576.0
    os.environ["NMF_FAULT_ARENA_LIMIT_MB"] = str(4.0 * M * N / 1048576.0 + 9 * pair_mb)     # X + room for nine pairs' factors at most
    try:
        W2, H2 = [ng.Matrix(w.copy(order="F")) for w in Ws], [ng.Matrix(h.copy(order="F")) for h in Hs]
        b2, kl2 = ng.update_div_restarts(W2, H2, ng.Matrix(X), max_iter=40, n_devices=1)
        os.environ["NMF_FAULT_ARENA_LIMIT_MB"] = "0.001"                                        # not even one pair: the error is named
        with pytest.raises(ng.NmfError) as e:
            ng.update_div_restarts(W2, H2, ng.Matrix(X), max_iter=2, n_devices=1)
        assert "does not fit" in str(e.value)
    finally:
        os.environ.pop("NMF_FAULT_ARENA_LIMIT_MB", None)
    assert b2 == b1 and kl2 == kl1 and all(np.array_equal(a.mat, b.mat) for a, b in zip(W1 + H1, W2 + H2))


def test_restart_workers_refuse_what_they_cannot_do(ng, oracle):
    M, N, K = 128, 256, 32
    X, _, _ = oracle.gen_problem(M, N, K, seed=43)
    Ws, Hs = _pairs(M, N, K, 3, 44)
    with pytest.raises(ng.NmfError) as e:      # more workers than devices, no explicit list
        ng.update_div_restarts([ng.Matrix(w) for w in Ws], [ng.Matrix(h) for h in Hs], ng.Matrix(X), max_iter=2, n_devices=ng.device_count() + 1)
    assert e.value.status == 1
    with pytest.raises(ng.NmfError) as e:      # a device that does not exist
        ng.update_div_restarts([ng.Matrix(w) for w in Ws], [ng.Matrix(h) for h in Hs], ng.Matrix(X), max_iter=2, n_devices=2, devices=[0, 99])
    assert e.value.status == 1


def test_restarts_main_example_runs_two_workers_through_the_c_abi(oracle, tmp_path):
    """examples/restarts_main.cpp: a plain C++ main, built with g++ against the header and the .so,
    runs R restarts of the reference's workflow over two workers; winner and factors against the oracle run on the same files"""
    exe = tmp_path / "restarts_main"
    pkg = os.path.join(ROOT, "nmf-gpu_amd")
    subprocess.run(["g++", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "restarts_main.cpp"),
                    "-L", pkg, "-lnmf_mi355x", f"-Wl,-rpath,{pkg}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], check=True)
    M, N, K, R = 512, 350, 64, 4
    X, _, _ = oracle.gen_problem(M, N, K, seed=45)
    Ws, Hs = _pairs(M, N, K, R, 46)
    oracle.write_bin(str(tmp_path / "X.bin"), X)
    for i in range(R):
        oracle.write_bin(str(tmp_path / f"W{i}.bin"), Ws[i])
        oracle.write_bin(str(tmp_path / f"H{i}.bin"), Hs[i])
    r = subprocess.run([str(exe), str(tmp_path), str(R), "2", "0", "0"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    lines = r.stdout.split("\n")
    best = int(lines[0].split()[1])
    kls = [float(l.split()[3]) for l in lines[1:1 + R]]
    ref = [oracle.update_div(Ws[i], Hs[i], X, 0.0, 200, 25) for i in range(R)]
    ref_kl = [oracle.kl_div(oracle.clamp(X), oracle.clamp(oracle.sgemm("nn", w, h))) for (w, h, _, _) in ref]
    assert np.allclose(kls, ref_kl, rtol=1e-4) and best == int(np.argmin(kls))
    for i in range(R):
        assert oracle.relF(oracle.read_bin(str(tmp_path / f"Wout{i}.bin")), ref[i][0]) < 1e-4
        assert oracle.relF(oracle.read_bin(str(tmp_path / f"Hout{i}.bin")), ref[i][1]) < 1e-4


def test_concurrent_ranks_each_with_an_rccl_communicator_capture_and_replay_their_graphs(ng, oracle):
    """What a one-GPU box can rehearse of the N > 1 run that no hardware has executed yet (round-3 VERDICT next 2c): several host
    threads at once, each inside the multi-device driver with its OWN one-rank RCCL communicator on device 0 -- ncclCommInitAll,
    the eager warm-up all-reduce, thread-local capture of the 32 / 8 / 1-iteration hipGraphs WITH the in-graph all-reduce, replay,
    all-reduced KL checks, pooled streams -- concurrently, on different problems (a split-kernel shape, a 64-column shape, a
    non-power-of-two K).  Every thread's result must equal, bit for bit, the same call made alone afterwards, and match the
    oracle.  What this cannot contain is RCCL between devices: N > 1 stays unmeasured and opt-in."""
    import threading
    shapes = [(1024, 4096, 64), (512, 2048, 100), (1024, 2048, 256), (768, 3000, 48)]
    probs = [oracle.gen_problem(M, N, K, seed=31 + i) for i, (M, N, K) in enumerate(shapes)]
    kw = dict(max_iter=80, n_devices=1, devices=[0], converge_thresh=1e-30, iter_check=40, use_graph=1)

    def run(i, out):
        X, W, H = probs[i]
        Wm, Hm = ng.Matrix(W.copy()), ng.Matrix(H.copy())
        r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), **kw)
        out[i] = (Wm.mat.copy(), Hm.mat.copy(), r)

    together, alone = {}, {}
    th = [threading.Thread(target=run, args=(i, together)) for i in range(len(shapes))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert len(together) == len(shapes)          # no thread died in the driver
    # a wait makes no HIP call while it waits and its closing synchronize was never refused because of a capture elsewhere
    # (round-4 VERDICT weak 2: the poll used to go through hipStreamQuery and swallowed up to 200 errors of any kind)
    assert ng.comm_capture_refusals() == 0
    for i in range(len(shapes)):
        run(i, alone)
    for i, (M, N, K) in enumerate(shapes):
        Wt, Ht, rt = together[i]
        Wa, Ha, ra = alone[i]
        assert rt["n_shards"] == 1 and rt["iterations"] == 80 and len(rt["kl"]) == 3 and rt["w_replicas_identical"] == 1
        assert np.array_equal(Wt, Wa) and np.array_equal(Ht, Ha) and rt["kl"] == ra["kl"], (M, N, K)
        X, W, H = probs[i]
        Wr, Hr, _, klr = oracle.update_div(W, H, X, 1e-30, 80, 40)
        assert oracle.relF(Wt, Wr) < 1e-5 and oracle.relF(Ht, Hr) < 1e-5 and np.allclose(rt["kl"], klr, rtol=2e-5), (M, N, K)


@pytest.mark.parametrize("K,G", [(160, 2), (100, 3), (12, 2)])
def test_sharded_64_column_kernel_at_a_rank_between_the_powers_of_two(ng, oracle, K, G):
    """The in-library driver with emulated shards where every rank runs the 64-column kernel (split_kernel = -1) at a K that is not a
    power of two: K = 160 (KT = 10, factors padded to 160) and K = 100 (KT = 7 with three steps of product 1 trimmed, factors padded
    to 128: the all-reduce operand and the slabs carry zero padding rows) -- and K = 12 on the K = 16 instantiation (factors padded to
    32: half of every slab is padding).  60 iterations against the oracle, replicas identical."""
    M, N = 512, 2048
    X, W, H = oracle.gen_problem(M, N, K, seed=41)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=60, emulate_shards=G, split_kernel=-1)
    assert r["n_shards"] == G and r["w_replicas_identical"] == 1 and r["iterations"] == 60
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 60, 25)
    assert oracle.relF(Wm.mat, Wr) < 1e-5 and oracle.relF(Hm.mat, Hr) < 1e-5


def test_probe_all_reduce_of_a_communicator(ng):
    """nmf_comm_probe: the communicator's first collective, with a deadline and a checked result -- what bench.py's negotiation runs on
    every rank before it commits to the in-library all-reduce (a one-rank communicator here; every rank must call it)."""
    c = ng.Comm(ng.Comm.unique_id(), 0, 1)
    try:
        c.probe()
        c.probe(5.0)
    finally:
        c.close()
