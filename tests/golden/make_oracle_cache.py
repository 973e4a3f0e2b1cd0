"""Write tests/golden/oracle_cache/*.npz: the results of the CPU-oracle computations of the GPU tests that are pure functions
of seeds (tests/util.py oracle_cache).  Runs the very functions the tests run -- nothing of the reference, no GPU:

    DITREE_WRITE_ORACLE_CACHE=1 python tests/golden/make_oracle_cache.py

Every test that reads a cache file re-computes a small probe live and fails when the two differ, so a cache that has gone
stale (oracle code or seeds changed) cannot stand in silently: re-run this script."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REPO)
os.environ["DITREE_WRITE_ORACLE_CACHE"] = "1"
os.environ["DITREE_ORACLE_CACHE"] = "0"          # compute, do not read back


def main():
    from tests import util
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    import tests.test_gpu_round_precision as RP
    onet = RP.make_net()
    for name in RP.WORKLOADS:
        util.oracle_cache(f"round_precision_{name}", lambda: RP.oracle_refs(onet, name))
        print("round_precision", name)
    import tests.test_gpu_scenarios as SC
    onet2 = SC.make_onet()
    for tag, H in (("race", 64), ("boxes", 64), ("rlarge2", 64), ("boxes", 32)):
        util.oracle_cache(f"scenario_{tag}_H{H}", lambda: SC.oracle_plan(onet2, tag, H))
        print("scenario", tag, H)
    import tests.test_gpu_ant_round as AR
    util.oracle_cache("ant_denoiser_rounds", lambda: AR.oracle_ant_denoiser_rounds(AR.make_ant_net()))
    print("ant_denoiser_rounds")
    for f in sorted(os.listdir(util.CACHE_DIR)):
        print(f, os.path.getsize(os.path.join(util.CACHE_DIR, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
