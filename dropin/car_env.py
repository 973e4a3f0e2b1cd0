from ditreeonlineplanner_amd.car_env import CarEnv  # noqa: F401
