"""Build libditree_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python -m ditreeonlineplanner_amd.build            # incremental
    python -m ditreeonlineplanner_amd.build --force
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libditree_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# (source, extra flags).  The geometry unit must not contract a*b+c (bit-exact flags).
UNITS = [
    ("geom_kernels.hip", ["-ffp-contract=off"]),
    ("ditree_api.hip", []),
    ("denoise_kernels.hip", []),
    ("denoise_host.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-DNDEBUG"]


def _newer(a: str, b: str) -> bool:
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force: bool = False, verbose: bool = True) -> str:
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(os.path.dirname(HERE), "include", "ditree.h"))
    objs = []
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        stale = force or _newer(s, o) or any(_newer(h, o) for h in hdrs)
        if stale:
            cmd = [HIPCC, *COMMON, *extra, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    if force or any(_newer(o, LIB) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
