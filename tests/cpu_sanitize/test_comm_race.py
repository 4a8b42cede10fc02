"""The abort of a communicator group against ranks on their way into / inside ncclAllReduce, under ThreadSanitizer on the CPU
(round-3 VERDICT weak 5 / next 2b).  nmf_comm_abort frees every RCCL handle of an ncclCommInitAll group, usually from a thread
other than the ones using them; until round 4 a rank read its handle without a lock on its way into ncclAllReduce, so the abort
could free it in between (a use-after-abort on the failure path).  Now a user holds the communicator's call mutex from its
abort check to the return of its RCCL call, and the aborter takes the handle out first and waits (briefly) for that mutex.

The product's nmf_comm.cpp is compiled with -fsanitize=thread and linked into tests/cpu_sanitize/comm_race_driver.cpp; RCCL is played
by tests/cpu_sanitize/fake_rccl.c (our own test double under RCCL's soname, preloaded: nmf_comm binds to an RCCL the process has
already mapped), which touches a heap object per communicator inside ncclAllReduce and frees it in ncclCommAbort -- a handle used
while or after it is freed is a ThreadSanitizer report and a non-zero exit.  No GPU and no HIP call are involved.
(The same driver built against round 3's nmf_comm.cpp exits with TSan's data-race report after a few rounds.)

CPU only: this directory is listed in .gpurunignore and never travels to the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
HERE = os.path.dirname(os.path.abspath(__file__))
CLANG = "/opt/rocm/lib/llvm/bin/clang"
CLANGXX = "/opt/rocm/lib/llvm/bin/clang++"
# A host-only build: nmf_comm.cpp is plain C++ over the HIP *API* header (its one kernel lives in nmf_kernels.hip and is stubbed
# by the driver), so there is no device pass, no GPU target and no GPU code object in anything built here.
HOST = ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include"]
TSAN = "-fsanitize=thread"


@pytest.mark.skipif(not (os.path.exists(CLANGXX) and os.path.exists(CLANG)), reason="needs ROCm's clang with the TSan runtime")
def test_abort_never_frees_a_communicator_under_a_rank_that_is_using_it(tmp_path):
    d = str(tmp_path)
    quiet = dict(stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    fake = os.path.join(d, "librccl.so.1")
    subprocess.run([CLANG, "-shared", "-fPIC", "-O1", "-g", TSAN, "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-Wl,-soname,librccl.so.1",
                    os.path.join(HERE, "fake_rccl.c"), "-o", fake], check=True, **quiet)
    objs = []
    for src in (os.path.join(ROOT, "nmf-gpu_amd", "csrc", "nmf_comm.cpp"), os.path.join(HERE, "comm_race_driver.cpp")):
        obj = os.path.join(d, os.path.basename(src) + ".o")
        r = subprocess.run([CLANGXX, "-O1", "-g", "-std=c++17", "-fPIC", TSAN] + HOST + ["-c", src, "-o", obj], **quiet)
        assert r.returncode == 0, r.stderr[-4000:]
        objs.append(obj)
    exe = os.path.join(d, "comm_race_driver")
    r = subprocess.run([CLANGXX, TSAN] + objs + ["-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-ldl", "-lpthread"], **quiet)
    assert r.returncode == 0, r.stderr[-4000:]
    env = dict(os.environ, LD_PRELOAD=fake, TSAN_OPTIONS="halt_on_error=1 exitcode=66 symbolize=0")
    r = subprocess.run([exe, "150"], env=env, timeout=300, **quiet)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    assert "150 rounds" in r.stdout and "ended by the abort" in r.stdout, r.stdout
