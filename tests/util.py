import functools
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, "ditreeonlineplanner_amd", "data")


def load_maze(name):
    return np.loadtxt(os.path.join(DATA, f"{name}.csv"), delimiter=",")


@functools.lru_cache(maxsize=None)
def golden(name):
    return dict(np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"), allow_pickle=False))


# ---------------------------------------------------------------------- cached oracle results
# Some GPU tests compare the engine with an ORACLE computation that is a pure function of seeds (a round of 1024 candidates
# through the torch-CPU fp32 network takes half a minute of host time).  Those results are kept under tests/golden/oracle_cache/
# (written by `python tests/golden/make_oracle_cache.py`, which runs the very functions the tests would run) so that the GPU suite
# spends its time on the GPU.  Every user re-computes a small PROBE of the cached quantity live and compares: a cache that no
# longer matches the oracle code fails the test instead of silently standing in for it.  No cache file -> computed on the spot.
CACHE_DIR = os.path.join(REPO, "tests", "golden", "oracle_cache")


def _flatten(obj, prefix, out):
    if isinstance(obj, dict):
        out[prefix + "__keys"] = np.array(sorted(obj.keys()))
        for k in obj:
            _flatten(obj[k], f"{prefix}{k}.", out)
    elif isinstance(obj, (list, tuple)):
        out[prefix + "__len"] = np.array(len(obj))
        for i, v in enumerate(obj):
            _flatten(v, f"{prefix}{i}.", out)
    elif obj is None:
        out[prefix + "__none"] = np.array(1)
    else:
        out[prefix + "v"] = np.asarray(obj)


def _unflatten(z, prefix):
    if prefix + "__keys" in z:
        return {str(k): _unflatten(z, f"{prefix}{k}.") for k in z[prefix + "__keys"]}
    if prefix + "__len" in z:
        return [_unflatten(z, f"{prefix}{i}.") for i in range(int(z[prefix + "__len"]))]
    if prefix + "__none" in z:
        return None
    v = z[prefix + "v"]
    return v[()] if v.ndim == 0 else v


def oracle_cache(name, compute):
    """``compute()`` -> nested dicts / lists of arrays and scalars; loaded from the cache file when there is one."""
    path = os.path.join(CACHE_DIR, name + ".npz")
    if os.path.exists(path) and os.environ.get("DITREE_ORACLE_CACHE", "1") != "0":
        with np.load(path, allow_pickle=False) as z:
            return _unflatten(dict(z), ""), True
    obj = compute()
    if os.environ.get("DITREE_WRITE_ORACLE_CACHE", "0") == "1":
        os.makedirs(CACHE_DIR, exist_ok=True)
        flat = {}
        _flatten(obj, "", flat)
        np.savez_compressed(path, **flat)
    return obj, False
