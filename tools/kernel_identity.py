"""Identity of the code of the half-step kernels: sha256 (first 16 hex digits) over the sources and build flags that determine the generated
code of the 64-column kernel family -- what the committed PMC passes (profiles/pmc_static.json) were taken of.  bench.py reports
`roofline.traffic` from those passes only when the running tree's identity equals the one stored with them (round-4 ADVICE: after a
kernel change the bytes per launch could be stale next to live timings)."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("nmf_fused16_impl.h", "nmf_fused16_inst.hip", "nmf_device.h", "nmf_kernels.h", "Makefile")


def kernel_sources_sha16(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        with open(os.path.join(root, "nmf-gpu_amd", "csrc", f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read() + b"\0")
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_sources_sha16())
