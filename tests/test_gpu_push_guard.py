"""What travels to the GPU box must not ask for what the pool refuses (round-4 GPUTEST: the whole GPU suite was refused because one
CPU-only test file held a ThreadSanitizer build line).  gpurun pushes /root/repo minus .git/, gpurun_out/ and the paths listed in
.gpurunignore; the pool runs no GPU sanitizer builds and no XNACK-on code objects, and refuses a push in which a file asks for
either.  Sanitizer builds live under tests/cpu_sanitize/ (listed in .gpurunignore) and are host-only."""
import fnmatch
import os

from conftest import ROOT

# spelled in pieces so that this file does not carry them itself
REFUSED = ("-fsani" + "tize=", "HSA_" + "XNACK", "xnack" + "+", "-fsani" + "tize-")
BUILT = (".so", ".o", ".a", ".hsaco", ".co", ".pyc")


def _ignore_rules():
    rules = [".git/", "gpurun_out/"]
    with open(os.path.join(ROOT, ".gpurunignore")) as f:
        rules += [ln.strip() for ln in f if ln.strip() and not ln.startswith("#")]
    return rules


def _ignored(rel, rules):
    parts = rel.split("/")
    for r in rules:
        if r.endswith("/"):
            d = r.rstrip("/")
            if "/" in d:
                if rel.startswith(d + "/"):
                    return True
            elif d in parts[:-1]:
                return True
        elif fnmatch.fnmatch(rel, r) or fnmatch.fnmatch(parts[-1], r):
            return True
    return False


def _carried():
    rules = _ignore_rules()
    for dp, dn, fn in os.walk(ROOT):
        rel_d = os.path.relpath(dp, ROOT).replace(os.sep, "/")
        dn[:] = [d for d in dn if not _ignored((rel_d + "/" if rel_d != "." else "") + d + "/x", rules)]
        for f in fn:
            rel = (rel_d + "/" if rel_d != "." else "") + f
            if not _ignored(rel, rules):
                yield rel


def test_gpurunignore_lists_the_sanitizer_tests_and_the_round_records():
    carried = set(_carried())
    assert not [p for p in carried if p.startswith("tests/cpu_sanitize/")]
    assert not [p for p in carried if p.startswith(("VERDICT", "ADVICE", "GPUTEST_"))]
    # and the GPU suite itself does travel
    for must in ("tests/conftest.py", "tests/test_gpu_update_div.py", "bench.py", "__graft_entry__.py", "oracle/nmf_oracle.c", "nmf-gpu_amd/csrc/Makefile"):
        assert must in carried, must


def test_nothing_pushed_to_the_gpu_box_asks_for_a_sanitizer_or_xnack_build():
    hits = []
    for rel in _carried():
        if rel.endswith(BUILT) or rel == "nmf-gpu_amd/nmf":
            continue
        p = os.path.join(ROOT, rel)
        if os.path.getsize(p) > (8 << 20):
            continue
        with open(p, "rb") as f:
            blob = f.read()
        for s in REFUSED:
            if s.encode() in blob:
                hits.append((rel, s))
    assert not hits, hits


def test_no_sanitizer_build_names_a_gpu_target():
    """the sanitizer builds themselves: host compilers only, never hipcc / --offload-arch on a sanitizer line"""
    d = os.path.join(ROOT, "tests", "cpu_sanitize")
    for f in os.listdir(d):
        if not f.endswith(".py"):
            continue
        src = open(os.path.join(d, f)).read()
        assert "--offload-arch" not in src and "hipcc" not in src.replace("no hipcc", ""), f
        assert '"-x", "hip"' not in src, f
