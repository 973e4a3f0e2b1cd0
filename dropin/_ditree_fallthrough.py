"""Fall-through for the drop-in packages.

`dropin/` holds packages with the reference's names (`common`, `planners`, `policies`, `lidar_sim`, `MPPI`) that
override only the modules the MI355X engine replaces.  Every other module of those packages (`planners.MPC`,
`policies.PD_controller`, `common.se3_utils`, ...) must keep resolving to the user's reference checkout, so each
package appends the same-named directory of that checkout to its `__path__`.  The checkout is looked for in
$DITREE_REFERENCE_ROOT (set by `python -m ditreeonlineplanner_amd.run`), the current directory and the later
`sys.path` entries."""
import os
import sys


def fall_through(pkg_path, pkg_name):
    here = [os.path.realpath(p) for p in pkg_path]
    roots = []
    env = os.environ.get("DITREE_REFERENCE_ROOT")
    if env:
        roots.append(env)
    roots.append(os.getcwd())
    roots.extend(p for p in sys.path if p)
    out = list(pkg_path)
    for r in roots:
        cand = os.path.join(r, *pkg_name.split("."))
        if os.path.isdir(cand) and os.path.realpath(cand) not in here and cand not in out:
            out.append(cand)
    return out
