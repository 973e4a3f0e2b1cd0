"""Reference import path `common.map_utils` -> the engine's module (the SAME module object, so that the drivers'
`common.map_utils.cc_calls = 0` / `+= common.map_utils.cc_calls` (run_scenarios.py:338,343) see the engine's count)."""
import sys

import ditreeonlineplanner_amd.common.map_utils as _m

sys.modules[__name__] = _m
