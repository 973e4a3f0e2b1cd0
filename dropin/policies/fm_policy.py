from ditreeonlineplanner_amd.policies.fm_policy import DiffusionSampler, load_metadata  # noqa: F401
