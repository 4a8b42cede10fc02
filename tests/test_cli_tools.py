"""The CLI's CPU-only tools: `nmf generate` restates matrix_export.py byte for byte, `nmf compare` is
test_output.sh with a tolerance.  Neither touches the GPU."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT
from test_oracle_golden import REF_MD5

CLI = os.path.join(ROOT, "nmf-gpu_amd", "nmf")


def _run(*args, cwd=None):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120, cwd=cwd)


def test_generate_reproduces_the_reference_inputs(tmp_path):
    """default shape and seed = matrix_export.py:4-7: the md5s of the files that script writes"""
    r = _run("generate", cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    for name, md5 in REF_MD5.items():
        assert hashlib.md5((tmp_path / f"{name}.bin").read_bytes()).hexdigest() == md5


def test_generate_other_shapes_match_the_oracle_generator(oracle, tmp_path):
    M, N, K, seed = 37, 53, 11, 5
    r = _run("generate", "--M", str(M), "--N", str(N), "--K", str(K), "--seed", str(seed),
             "--X", str(tmp_path / "x.bin"), "--W", str(tmp_path / "w.bin"), "--H", str(tmp_path / "h.bin"))
    assert r.returncode == 0, r.stderr
    X, W, H = oracle.gen_problem(M, N, K, seed=seed)
    for f, A in (("x.bin", X), ("w.bin", W), ("h.bin", H)):
        assert np.array_equal(oracle.read_bin(str(tmp_path / f)), A)
    assert _run("generate", "--M", "0").returncode != 0


def test_compare_is_test_output_with_a_tolerance(oracle, tmp_path):
    A = np.asfortranarray(np.random.default_rng(0).random((40, 30), dtype=np.float32))
    B = A * np.float32(1 + 3e-5)
    oracle.write_bin(str(tmp_path / "a.bin"), A); oracle.write_bin(str(tmp_path / "b.bin"), B)
    r = _run("compare", str(tmp_path / "a.bin"), str(tmp_path / "a.bin"))
    assert r.returncode == 0 and "bytes identical" in r.stdout
    r = _run("compare", str(tmp_path / "b.bin"), str(tmp_path / "a.bin"))
    assert r.returncode == 0 and "bytes differ" in r.stdout and "matches" in r.stdout          # 3e-5 <= 1e-4
    r = _run("compare", str(tmp_path / "b.bin"), str(tmp_path / "a.bin"), "--tol", "1e-6")
    assert r.returncode == 1 and "differs" in r.stdout
    oracle.write_bin(str(tmp_path / "c.bin"), np.asfortranarray(A[:, :7]))
    assert _run("compare", str(tmp_path / "c.bin"), str(tmp_path / "a.bin")).returncode == 1  # shapes differ
    r = _run("compare", str(tmp_path / "nope.bin"), str(tmp_path / "a.bin"))
    assert r.returncode != 0 and "cannot open" in r.stderr
    # the reference's golden pair against itself, and against each other's inputs
    assert _run("compare", os.path.join(GOLDEN, "Wtest.bin"), os.path.join(GOLDEN, "Wtest.bin")).returncode == 0


def _build_reference_main(tmp_path):
    exe = tmp_path / "bin" / "nmf_ref_main"
    exe.parent.mkdir()
    pkg = os.path.join(ROOT, "nmf-gpu_amd")
    subprocess.run(["g++", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "reference_main.cpp"),
                    "-L", pkg, "-lnmf_mi355x", f"-Wl,-rpath,{pkg}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_reference_main_rebound_compiles_and_links(tmp_path):
    """the reference's main() with update_div in place of run_async builds with the host
    compiler against the public header and the shared library alone (no HIP, no torch)."""
    exe = _build_reference_main(tmp_path)
    r = subprocess.run([str(exe)], capture_output=True, text=True, cwd=exe.parent, timeout=60)   # no ../X.bin here
    assert r.returncode == 1


@pytest.mark.gpu
def test_reference_main_rebound_runs_the_reference_workflow(oracle, tmp_path):
    """matrix_export.py -> ./nmf -> test_output.sh with the rebound main: ../X.bin ../W.bin ../H.bin in, 200 iterations,
    ../Wout.bin ../Hout.bin out (cuda/nmf.cu:30-51), checked against the oracle run on the same files."""
    exe = _build_reference_main(tmp_path)
    assert _run("generate", "--M", "512", "--N", "350", "--K", "64", cwd=tmp_path).returncode == 0
    r = subprocess.run([str(exe)], capture_output=True, text=True, cwd=exe.parent, timeout=300)
    assert r.returncode == 0, r.stderr
    X, W, H = (oracle.read_bin(str(tmp_path / f)) for f in ("X.bin", "W.bin", "H.bin"))
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 200, 25)
    assert oracle.relF(oracle.read_bin(str(tmp_path / "Wout.bin")), Wr) < 1e-4
    assert oracle.relF(oracle.read_bin(str(tmp_path / "Hout.bin")), Hr) < 1e-4


def _build_example(tmp_path, name):
    exe = tmp_path / "bin" / name
    exe.parent.mkdir(exist_ok=True)
    pkg = os.path.join(ROOT, "nmf-gpu_amd")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", name + ".cpp"),
                    "-L", pkg, "-lnmf_mi355x", f"-Wl,-rpath,{pkg}", "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)], check=True)
    return exe


def test_snapshot_main_on_the_cxx_surface_compiles_and_links(tmp_path):
    """include/nmf_mi355x.hpp -- the snapshot's own C++ surface (class Matrix, matrix_multiply ..., read_matrix, run_async:
    cuda/matrix.cuh:18-52, cuda/nmf.cu:13-28) as a header-only layer over the C ABI -- and the snapshot's main() written on it
    (examples/snapshot_main.cpp) build with the host compiler alone; without ../X.bin the program ends like the reference's
    read_matrix would if it checked fopen: a message and a non-zero exit (error-check.hpp:12-17)."""
    exe = _build_example(tmp_path, "snapshot_main")
    r = subprocess.run([str(exe)], capture_output=True, text=True, cwd=exe.parent, timeout=60)
    assert r.returncode != 0 and "read_matrix" in r.stderr and "cannot open" in r.stderr
    # the header is plain C++17: clean under -Wall -Wextra -pedantic with both host compilers
    for cxx in ("g++", "/opt/rocm/lib/llvm/bin/clang++"):
        if cxx.startswith("/") and not os.path.exists(cxx):
            continue
        subprocess.run([cxx, "-std=c++17", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "examples", "snapshot_main.cpp")], check=True)


@pytest.mark.gpu
def test_snapshot_main_runs_and_its_operator_iteration_agrees_with_the_fused_loop(oracle, tmp_path):
    """'may additionally offer a C++ Matrix-like RAII wrapper mirroring matrix.cuh:18-39'.  (i) The snapshot's main on
    that surface -- read_matrix x 3, run_async, write_matrix x 2 (cuda/nmf.cu:30-51) -- against the oracle after its 200
    iterations; (ii) the reference's own iteration, update_h / update_w as sixteen calls of the mirrored operators
    (cuda/nmf.cu:118-176: matrix_multiply, set_epsilon, element_divide, sum_cols, matrix_multiply_AtB, col_divide,
    element_multiply; ... sum_rows, matrix_multiply_ABt, row_divide ...), 20 iterations on the same files: against the oracle,
    and against run_async's fused loop (the two differ in summation order only)."""
    exe = _build_example(tmp_path, "snapshot_main")
    assert _run("generate", "--M", "700", "--N", "450", "--K", "96", cwd=tmp_path).returncode == 0
    X, W, H = (oracle.read_bin(str(tmp_path / f)) for f in ("X.bin", "W.bin", "H.bin"))
    r = subprocess.run([str(exe)], capture_output=True, text=True, cwd=exe.parent, timeout=300)
    assert r.returncode == 0, r.stderr
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 200, 25)
    assert oracle.relF(oracle.read_bin(str(tmp_path / "Wout.bin")), Wr) < 1e-5
    assert oracle.relF(oracle.read_bin(str(tmp_path / "Hout.bin")), Hr) < 1e-5
    r = subprocess.run([str(exe), "operators", "20"], capture_output=True, text=True, cwd=exe.parent, timeout=300)
    assert r.returncode == 0, r.stderr
    Wo, Ho = oracle.read_bin(str(tmp_path / "Wout.bin")), oracle.read_bin(str(tmp_path / "Hout.bin"))
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 20, 25)
    assert oracle.relF(Wo, Wr) < 5e-6 and oracle.relF(Ho, Hr) < 5e-6
    import nmf_gpu_amd as ng
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=20)
    assert oracle.relF(Wo, Wm.mat) < 5e-6 and oracle.relF(Ho, Hm.mat) < 5e-6


def test_sharded_main_compiles_and_links(tmp_path):
    exe = _build_example(tmp_path, "sharded_main")
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "usage" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,G", [(1024, 4096, 64, 2), (1024, 4096, 64, 3), (512, 700, 256, 4), (512, 3000, 100, 3)])
def test_in_library_multi_device_driver_with_emulated_shards(oracle, tmp_path, M, N, K, G):
    """multi-GPU is callee-internal.  A plain C++ main (examples/sharded_main.cpp, built with g++ against the
    header and the .so) calls update_div_ex once on host matrices; the library shards the columns over G ranks -- one host
    thread, one solver, one stream each -- all-reduces [Z*H' ; rowsum(H)] every iteration and gathers H.  Here the G ranks
    share the one GPU of the box (nmf_opts.emulate_shards: the RCCL call replaced by a rank-ordered device sum); the driver,
    the per-rank loop and the solver's sharded W-step are the production code.  200 iterations against the oracle at 1e-5
    (cfg2 shape: BASELINE config 2), replicas of W bit-identical, and G = 1-vs-G within the all-reduce's reordering.
    K = 100 (split kernel per rank, computing on 112 in factors padded to 128: the all-reduce operand carries zero padding rows)
    is one of round 4's ranks between the powers of two; the 64-column kernel under sharding at such a rank:
    tests/test_gpu_multi.py::test_sharded_64_column_kernel_at_a_rank_between_the_powers_of_two."""
    exe = _build_example(tmp_path, "sharded_main")
    assert _run("generate", "--M", str(M), "--N", str(N), "--K", str(K), cwd=tmp_path).returncode == 0
    r = subprocess.run([str(exe), str(tmp_path), str(G), "emulate"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    f = r.stdout.split()
    assert f[:6] == ["shards", str(G), "iterations", "200", "w_replicas_identical", "1"], r.stdout
    X, W, H = (oracle.read_bin(str(tmp_path / f)) for f in ("X.bin", "W.bin", "H.bin"))
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 200, 25)
    eW = oracle.relF(oracle.read_bin(str(tmp_path / "Wout.bin")), Wr)
    eH = oracle.relF(oracle.read_bin(str(tmp_path / "Hout.bin")), Hr)
    assert eW < 1e-5 and eH < 1e-5, (eW, eH)


@pytest.mark.gpu
def test_in_library_driver_over_a_real_rccl_communicator_of_one_rank(ng, oracle):
    """the same driver through ncclCommInitAll and the in-graph RCCL all-reduce, with the one device a test box has:
    n_devices = 1 takes the ordinary path, an explicit device list of length 1 is refused as pointless, so the RCCL leg is
    exercised through the solver-level communicator API the driver uses (nmf_comm_init_rank, in-graph all-reduce)"""
    M, N, K = 512, 2048, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=21)
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 20, 25)
    c = ng.Comm(ng.Comm.unique_id(), 0, 1)
    s = ng.Solver(M, N, K, comm=c)
    s.upload(W, H, X)
    s.iterate(20)
    Wg, Hg = s.download()
    s.close(); c.close()
    assert oracle.relF(Wg, Wr) < 1e-5 and oracle.relF(Hg, Hr) < 1e-5
    assert ng.lib().nmf_worth_sharding(4096, 262144, 256, 8) == 1 and ng.lib().nmf_worth_sharding(1024, 4096, 64, 8) == 0


@pytest.mark.gpu
def test_multi_device_requests_that_cannot_be_met_are_refused(ng):
    """n_devices beyond what is visible, shards without host data, more ranks than columns: a status, never a hang"""
    import numpy as np
    rng = np.random.default_rng(0)
    W, H, X = (ng.Matrix(rng.random(s, dtype=np.float32)) for s in ((64, 8), (8, 96), (64, 96)))
    with pytest.raises(ng.NmfError) as e:
        ng.update_div_ex(W, H, X, max_iter=2, n_devices=ng.device_count() + 1)
    assert e.value.status == 1
    with pytest.raises(ng.NmfError) as e:
        ng.update_div_ex(W, H, X, max_iter=2, emulate_shards=9)
    assert e.value.status == 1
    Hs, Xs = ng.Matrix(rng.random((8, 2), dtype=np.float32)), ng.Matrix(rng.random((64, 2), dtype=np.float32))
    with pytest.raises(ng.NmfError) as e:
        ng.update_div_ex(W, Hs, Xs, max_iter=2, emulate_shards=3)       # 2 columns over 3 ranks
    assert e.value.status == 1
    r = ng.update_div_ex(W, H, X, max_iter=2, n_devices=1)
    assert r["n_shards"] == 1 and r["w_replicas_identical"] == 1
    r = ng.update_div_ex(W, H, X, max_iter=3, emulate_shards=2, converge_thresh=1e-30, iter_check=1)
    assert r["n_shards"] == 2 and r["iterations"] == 3 and len(r["kl"]) == 4 and r["kl"][0] > r["kl"][-1]


@pytest.mark.gpu
def test_multi_device_driver_over_real_rccl_with_the_one_device_of_the_box(ng, oracle):
    """update_div_ex(n_devices=1, devices=[0]): the in-library driver end to end over RCCL itself -- ncclCommInitAll, the rank's
    host thread, the eager warm-up all-reduce, the all-reduce captured in the rank's hipGraphs, KL checks all-reduced, H gathered --
    with the single rank a one-GPU box allows (N > 1 ranks over xGMI remain unmeasured)"""
    import numpy as np
    M, N, K = 1024, 8192, 64
    X, W, H = oracle.gen_problem(M, N, K, seed=23)
    Wm, Hm = ng.Matrix(W), ng.Matrix(H)
    r = ng.update_div_ex(Wm, Hm, ng.Matrix(X), max_iter=40, n_devices=1, devices=[0], converge_thresh=1e-30, iter_check=20, use_graph=1)
    assert r["n_shards"] == 1 and r["w_replicas_identical"] == 1 and r["iterations"] == 40 and len(r["kl"]) == 3
    Wr, Hr, _, klr = oracle.update_div(W, H, X, 1e-30, 40, 20)
    assert oracle.relF(Wm.mat, Wr) < 1e-5 and oracle.relF(Hm.mat, Hr) < 1e-5 and np.allclose(r["kl"], klr, rtol=2e-5)


@pytest.mark.gpu
def test_cli_restarts_writes_the_best_of_r_initialisations(oracle, tmp_path):
    """`nmf --restarts r` (paper section 3.2 through the CLI): the W, H files are restart 0, r - 1 more pairs come from
    MT19937(seed + i); the pair with the lowest final KL is written.  Restart 0 must be what the plain run of the same files
    gives (to the batched split's summation order), and the written pair must be the arg-min of the printed KLs."""
    assert _run("generate", "--M", "512", "--N", "350", "--K", "64", cwd=tmp_path).returncode == 0
    r = _run("--X", "X.bin", "--W", "W.bin", "--H", "H.bin", "--Wout", "Wb.bin", "--Hout", "Hb.bin", "--iters", "60", "--restarts", "5", "--seed", "3", cwd=tmp_path)
    assert r.returncode == 0, r.stderr
    kls = [float(l.split()[3]) for l in r.stdout.split("\n") if l.startswith("restart ")]
    best = [i for i, l in enumerate(l for l in r.stdout.split("\n") if l.startswith("restart ")) if "<- best" in l]
    assert len(kls) == 5 and best == [int(np.argmin(kls))]
    X, W, H = (oracle.read_bin(str(tmp_path / f)) for f in ("X.bin", "W.bin", "H.bin"))
    Wr, Hr, _, _ = oracle.update_div(W, H, X, 0.0, 60, 25)
    kl0 = oracle.kl_div(oracle.clamp(X), oracle.clamp(oracle.sgemm("nn", Wr, Hr)))
    assert abs(kls[0] - kl0) <= 1e-4 * kl0
    Wb, Hb = oracle.read_bin(str(tmp_path / "Wb.bin")), oracle.read_bin(str(tmp_path / "Hb.bin"))
    klb = oracle.kl_div(oracle.clamp(X), oracle.clamp(oracle.sgemm("nn", Wb, Hb)))
    assert abs(klb - min(kls)) <= 1e-4 * klb
    if best == [0]:
        assert oracle.relF(Wb, Wr) < 1e-4
