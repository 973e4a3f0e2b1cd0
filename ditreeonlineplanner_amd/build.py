"""Build libditree_hip.so in-tree with hipcc for gfx950 (no cmake, no JIT cache).

    python -m ditreeonlineplanner_amd.build            # incremental
    python -m ditreeonlineplanner_amd.build --force

The library carries a BUILD ID = sha256 over csrc/*.hip, csrc/*.h, include/ditree.h and the compile flags
(`ditree_build_id()`, also greppable in the binary as DITREE_BUILD_ID=<hex>).  Staleness is decided by content, not by
modification times (a checkout or a copied snapshot does not keep them): every object file has a side file with the hash
of what it was compiled from, `build()` recompiles what does not match, and `_lib.lib()` refuses to load a library whose
ID differs from the sources next to it -- a stale shipped .so cannot pass silently.
"""
from __future__ import annotations

import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libditree_hip.so")
HEADER = os.path.join(os.path.dirname(HERE), "include", "ditree.h")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

# (source, extra flags).  The geometry unit must not contract a*b+c (bit-exact flags).
UNITS = [
    ("geom_kernels.hip", ["-ffp-contract=off"]),
    ("mppi_kernels.hip", ["-ffp-contract=off"]),
    ("ant_kernels.hip", ["-ffp-contract=off"]),
    ("mppi_ant_kernels.hip", ["-ffp-contract=off"]),
    ("ditree_api.hip", []),
    ("denoise_kernels.hip", []),
    ("denoise_host.hip", []),
]
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
          "-Wno-unused-variable", "-Wno-unused-value", "-Wno-unused-result", "-DNDEBUG"]
ID_UNIT = "ditree_api.hip"          # the unit that embeds the build id


def _read(path: str) -> bytes:
    with open(path, "rb") as f:
        return f.read()


def _headers():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [HEADER]


def source_id() -> str:
    """Build id of the sources as they are on disk now (16 hex digits)."""
    h = hashlib.sha256()
    for path in sorted(os.path.join(CSRC, u) for u, _ in UNITS) + _headers():
        h.update(os.path.basename(path).encode() + b"\0" + _read(path) + b"\0")
    h.update(" ".join(COMMON + [f for _, extra in UNITS for f in extra]).encode())
    return h.hexdigest()[:16]


def library_id(path: str = LIB):
    """Build id embedded in a built library (without loading it), or None."""
    if not os.path.exists(path):
        return None
    m = re.search(rb"DITREE_BUILD_ID=([0-9a-f]{16})", _read(path))
    return m.group(1).decode() if m else None


def _unit_hash(src: str, extra, build_id: str) -> str:
    h = hashlib.sha256(_read(os.path.join(CSRC, src)))
    for p in _headers():
        h.update(_read(p))
    h.update(" ".join(COMMON + list(extra)).encode())
    if src == ID_UNIT:
        h.update(build_id.encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    bid = source_id()
    objs = []
    relink = force or library_id() != bid
    for src, extra in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        side = o + ".srchash"
        objs.append(o)
        want = _unit_hash(src, extra, bid)
        have = _read(side).decode().strip() if os.path.exists(side) and os.path.exists(o) else None
        if force or have != want:
            defs = [f'-DDITREE_BUILD_ID_STR="{bid}"'] if src == ID_UNIT else []
            cmd = [HIPCC, *COMMON, *extra, *defs, "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            with open(side, "w") as f:
                f.write(want + "\n")
            relink = True
    if relink:
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    got = library_id()
    if got != bid:
        raise RuntimeError(f"built library carries id {got}, sources are {bid}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("build id", library_id())
