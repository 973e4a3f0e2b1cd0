from ditreeonlineplanner_amd.planners.RRT import *  # noqa: F401,F403
from ditreeonlineplanner_amd.planners.RRT import RRT_Planner  # noqa: F401
