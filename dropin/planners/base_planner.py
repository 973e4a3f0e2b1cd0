from ditreeonlineplanner_amd.planners.base_planner import *  # noqa: F401,F403
from ditreeonlineplanner_amd.planners.base_planner import BasePlanner, Node  # noqa: F401
