"""bench.py quotes `roofline.traffic` (HBM bytes per launch from the committed PMC passes, profiles/pmc_static.json) only while the passes
were taken of the kernel sources in this tree (tools/kernel_identity.py; round-4 ADVICE: stale bytes next to live timings).  This holds the
committed passes to the tree: after a change to the half-step kernel's sources, re-collect them (`ROUND=rNN bash tools/profile.sh pmc`)."""
import json
import os
import sys

from conftest import ROOT


def test_the_committed_pmc_passes_are_of_this_trees_kernel_sources():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from kernel_identity import kernel_sources_sha16
    with open(os.path.join(ROOT, "profiles", "pmc_static.json")) as f:
        e = json.load(f)["4096x65536x256"]
    assert e.get("kernel_sources_sha16") == kernel_sources_sha16(), "profiles/pmc_static.json is stale: bench.py would report traffic = null"
    assert e["hbm_bytes_per_launch"]["H"] > 1.2e9 and e["hbm_bytes_per_launch"]["W"] > 1.1e9      # >= the algorithmic 1.21 / 1.15 GB


def test_bench_reports_null_traffic_for_other_kernel_sources(tmp_path, monkeypatch):
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    e = bench.pmc_static(4096, 65536, 256)
    assert e is not None and not e.get("stale")
    import kernel_identity
    monkeypatch.setattr(kernel_identity, "kernel_sources_sha16", lambda root=None: "0123456789abcdef")
    e = bench.pmc_static(4096, 65536, 256)
    assert e is not None and "traffic not quoted" in e["stale"]
