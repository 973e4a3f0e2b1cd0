from ditreeonlineplanner_amd.common.fm_utils import get_timesteps  # noqa: F401
