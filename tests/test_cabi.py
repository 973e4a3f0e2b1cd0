"""CPU: the C-ABI library loads and exports every symbol include/ditree.h declares."""
import ctypes
import os
import re

from tests.util import REPO


def test_header_symbols_exported():
    from ditreeonlineplanner_amd import _lib
    hdr = open(os.path.join(REPO, "include", "ditree.h")).read()
    declared = set(re.findall(r"\b(ditree_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    h = _lib.lib()
    for name in declared:
        assert hasattr(h, name), name
    assert h.ditree_version() == 400


def test_no_oracle_import_in_product():
    pkg = os.path.join(REPO, "ditreeonlineplanner_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(root, f)


def test_product_fails_loudly_without_gpu():
    import pytest
    import torch
    from ditreeonlineplanner_amd import _lib
    from ditreeonlineplanner_amd.ops import Context
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DitreeLibraryError):
        Context()


def test_halo_kernel_loop_has_no_scratch_traffic(tmp_path):
    """The 16x16x32 halo kernel waits for its LDS-DMA stages with a *counted* s_waitcnt vmcnt(2) inside the K loop.
    That is only sound while the loop body issues no other vector-memory operation: a register spill reloaded or stored
    there (scratch_* counts in vmcnt) would change what the count means.  Guard the generated code, not the source."""
    import re
    import subprocess
    src = os.path.join(REPO, "ditreeonlineplanner_amd", "csrc", "denoise_kernels.hip")
    out = tmp_path / "dk.s"
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    subprocess.run([hipcc, "-S", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", src, "-o", str(out)],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    # every instantiation: <element type 0 bf16 / 1 f16, split 0 / 1>; for the split kernel also <SHORT (the L = 8 / 4 epilogue),
    # SPLITK (latency mode), NP (activation pieces: 5, or 6 at L = 4)>.  (The epilogue has a loop of its own -- the FiLM rows going to LDS -- so the chunk loop is the one
    # that holds the MFMAs.)  The default instantiations do not spill at all, which is asserted too (a spill in the epilogue
    # costs a store round trip per reload); the SHORT epilogue may spill a few registers OUTSIDE the chunk loop.
    cases = [(0, 0, "", "bf16"), (1, 0, "", "f16")]
    cases += [(et, 1, f"ELb{sh}ELb{sk}ELi5", mf) for et, mf in ((0, "bf16"), (1, "f16")) for sh in (0, 1) for sk in (0, 1)]
    cases += [(et, 1, "ELb1ELb0ELi6", mf) for et, mf in ((0, "bf16"), (1, "f16"))]      # six activation pieces (L = 4)
    for et, split, extra, mfma in cases:
        name = (f"_Z21conv3_halo16x3_kernelILi{et}{extra}EEv14ConvGemmParams:" if split
                else f"_Z19conv3_halo16_kernelILi{et}ELb0EEv14ConvGemmParams:")
        start = text.index(name)
        body = text[start:text.index(".Lfunc_end", start)].splitlines()
        headers = [n for n, l in enumerate(body) if "Loop Header" in l]
        loops = []
        for h in headers:
            label = body[h].split(":")[0].strip()
            back = [n for n, l in enumerate(body) if re.search(r"s_cbranch\w+\s+" + re.escape(label) + r"\b", l)]
            assert back, "no backward branch to a loop header"
            loops.append(body[h:back[-1] + 1])
        loops = [lp for lp in loops if any("v_mfma" in l for l in lp)]
        assert len(loops) == 1, "expected exactly one loop with MFMAs (the chunk loop) in the halo kernel"
        loop = loops[0]
        if extra in ("", "ELb0ELb0ELi5"):
            assert not any("scratch_" in l for l in body), (et, split, extra, "register spills in the halo kernel")
        assert sum(f"v_mfma_f32_16x16x32_{mfma}" in l for l in loop) == (288 if split else 192)   # 3 K-steps x 64 (96: hi/lo) MFMAs
        assert any(("vmcnt(3)" if extra.endswith("ELi6") else "vmcnt(2)") in l for l in loop)      # the counted wait of T = 1
        offenders = [l.strip() for l in loop if "scratch_" in l or re.search(r"\b(global|flat)_(load|store)", l)]
        assert not offenders, (et, split, extra, offenders)


def test_library_carries_the_build_id_of_its_sources(monkeypatch):
    """A stale libditree_hip.so (built from other sources than the ones next to it, e.g. shipped with a snapshot) must not
    load: the id is a content hash of csrc + include/ditree.h + flags, embedded in the binary and exported."""
    import pytest
    from ditreeonlineplanner_amd import _lib, build
    assert build.library_id() == build.source_id() == _lib.build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", _lib.build_id())
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(build, "source_id", lambda: "0123456789abcdef")
    with pytest.raises(_lib.DitreeLibraryError, match="stale"):
        _lib.lib()
