from ditreeonlineplanner_amd.policies.uniform_policy import UniformSampler  # noqa: F401
