"""`from MPPI.mppi import MPPI` (run_scenarios_with_lidar_MPPI.py:10): imported by the reference, not shipped by it."""
from ditreeonlineplanner_amd.mppi import MPPI  # noqa: F401
