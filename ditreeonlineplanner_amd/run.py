"""Run one of the reference's driver scripts, unchanged, on the MI355X engine:

    python -m ditreeonlineplanner_amd.run /path/to/DiTreeOnlinePlanner/run_scenarios.py [script args...]

`python script.py` puts the script's directory FIRST on sys.path, so the reference's own `planners/`, `policies/`,
`car_env.py` would win over any PYTHONPATH entry.  This launcher puts `dropin/` (engine-backed modules under the
reference's import paths) in front, the script's directory behind it (for everything the engine does not replace:
`planners.MPC`, `obstacle_insertion`, `plot_logger`, `cfgs/`, `maps/`, `metadata/`), changes into the script's
directory (the reference opens `metadata/{env_id}.pt`, `cfgs/*.yaml` relative to the CWD) and executes the script as
`__main__`."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    root = os.path.dirname(script)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    dropin = os.path.join(repo, "dropin")
    os.environ["DITREE_REFERENCE_ROOT"] = root
    sys.path[:] = [dropin, repo, root] + [p for p in sys.path if p not in ("", dropin, repo, root)]
    os.chdir(root)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
