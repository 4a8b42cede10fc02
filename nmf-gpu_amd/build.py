"""Build libnmf_mi355x.so (+ the nmf CLI) for gfx950 with hipcc via csrc/Makefile."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnmf_mi355x.so")
CLI_PATH = os.path.join(_HERE, "nmf")


def build(verbose: bool = False) -> str:
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "all"]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.run(cmd, check=True)
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("hipcc build did not produce " + LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(verbose=True))
