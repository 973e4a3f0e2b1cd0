"""`from drone_env import DroneEnv` (run_scenarios.py:26): the reference imports this module but does not ship it."""
from ditreeonlineplanner_amd.stubs import DroneEnv  # noqa: F401
