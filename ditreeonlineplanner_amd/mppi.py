"""`MPPI.mppi.MPPI` -- the controller `run_scenarios_with_lidar_MPPI.py:10,339-449` imports.  The reference repository
does not contain this module ("MPPI: module absent from the reference, no oracle exists at
all"), so there are no semantics to reproduce.  What BASELINE config 5 asks of the hot path -- 65 536 rollouts of T = 16
bicycle steps with collision / goal tests -- is the rollout kernel measured alone (`bench.py --workload rollout`).
The class exists so the script's import block resolves; constructing it says what is missing."""


class MPPI:
    def __init__(self, maze_data=None, T=16, K=10, nx=6, nu=2, **kw):
        raise NotImplementedError(
            "MPPI.mppi is imported by run_scenarios_with_lidar_MPPI.py but is not part of the reference repository: there "
            "is no cost function, sampling scheme or update rule to be faithful to.  The rollout workload it would drive is "
            "`ExpansionEngine` / `Context.car_rollout` (K x T bicycle steps per launch, bench.py --workload rollout).")
