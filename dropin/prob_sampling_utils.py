"""Reference import path `prob_sampling_utils` -> host-side sampling-probability maps of the engine package."""
from ditreeonlineplanner_amd.prob_sampling_utils import combine_log_blend, edt_prior, gaussian_map  # noqa: F401
