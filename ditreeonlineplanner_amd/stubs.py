"""Names the reference's scripts import at module top but the reference does not ship (`drone_env.DroneEnv`,
`planners.random_tree.RandomTreePlanner`; run_scenarios.py:26,40).  They exist so the scripts start; the car / DiTree
path never instantiates them, and instantiating one says so instead of falling back to anything."""


class _NotInReference:
    _what = ""

    def __init__(self, *a, **k):
        raise NotImplementedError(f"{type(self).__name__}: {self._what} is imported by the reference's scripts but is not "
                                  "part of the reference repository; the MI355X engine covers the car / DiTree path only")


class DroneEnv(_NotInReference):
    _what = "drone_env.py"


class RandomTreePlanner(_NotInReference):
    _what = "planners/random_tree.py"
