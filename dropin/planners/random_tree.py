"""`from planners.random_tree import RandomTreePlanner` (run_scenarios.py:40): imported by the reference, not shipped by it."""
from ditreeonlineplanner_amd.stubs import RandomTreePlanner  # noqa: F401
