import functools
import os

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(REPO, "ditreeonlineplanner_amd", "data")


def load_maze(name):
    return np.loadtxt(os.path.join(DATA, f"{name}.csv"), delimiter=",")


@functools.lru_cache(maxsize=None)
def golden(name):
    return dict(np.load(os.path.join(REPO, "tests", "golden", f"{name}.npz"), allow_pickle=False))
