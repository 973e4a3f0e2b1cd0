"""The samples of a whole round, drawn in bulk from the reference's two global RNG streams -- element for element what
``BasePlanner.random_node_sample`` / ``RRT_Planner.plan`` draw one candidate at a time (planners/base_planner.py:157-207,
planners/RRT.py:134-140,153-156), and leaving ``random`` and ``np.random`` in exactly the state the loop leaves them in.

Why it can be done: every draw of that loop is a fixed function of raw generator output.
  * ``random.random()``: one double of the Python Mersenne Twister per call (two words, formed as below).
  * ``np.random.uniform(lo, hi, size=(1, 1))``: legacy RandomState, ``lo + (hi - lo) * d`` with ``d`` one 53-bit double made of
    two consecutive 32-bit words, ``((w0 >> 5) * 2**26 + (w1 >> 6)) / 2**53``.
  * ``np.random.choice(n_cells, size=1, p=p)`` (run_type >= 2): ``cdf = p.cumsum(); cdf /= cdf[-1]``, one double ``d``,
    ``cdf.searchsorted(d, side='right')``.
  * ``np.random.choice(np.arange(n))`` (run_type >= 1 with a remaining reference path): legacy ``randint(0, n)`` = masked
    rejection on single 32-bit words, ``w & mask`` until it is <= n - 1 (mask = the smallest 2**k - 1 >= n - 1).
So the raw words are pulled once (``np.random.get_bit_generator().random_raw``), the per-candidate consumption is resolved on
the host, and the generators are then re-positioned behind exactly the words the loop would have consumed.
``tests/test_draw_round.py`` holds the equality (samples, conditioning goals, both generator states) for every run_type.
"""
from __future__ import annotations

import random

import numpy as np

_TWO26, _TWO53 = 67108864.0, 9007199254740992.0


def _doubles(words):
    """Legacy ``rk_double`` over consecutive word pairs: words (..., 2) uint64 -> float64."""
    return ((words[..., 0] >> np.uint64(5)).astype(np.float64) * _TWO26 + (words[..., 1] >> np.uint64(6)).astype(np.float64)) / _TWO53


class _Streams:
    """Snapshot of both global generators + a generous block of their raw output; ``commit`` re-positions them.

    Python's ``random`` is the same MT19937 as numpy's legacy generator and ``random.random()`` forms its double from two
    words exactly as ``rk_double`` does (``(a >> 5, b >> 6)``, CPython Modules/_randommodule.c), so its state is loaded into a
    numpy ``MT19937`` bit generator, the words are pulled in one call, and the advanced state is written back."""

    def __init__(self, n_py, n_words):
        self.py_state = random.getstate()
        self.np_state = np.random.get_state()
        self._bg = np.random.MT19937()
        self._load_py(self.py_state)
        self.py = _doubles(self._bg.random_raw(2 * n_py).reshape(n_py, 2)) if n_py else np.zeros(0)
        self.words = np.random.get_bit_generator().random_raw(n_words)          # uint64 array of 32-bit outputs
        self.n_py, self.n_words = n_py, n_words

    def _load_py(self, st):
        version, internal, _ = st
        if version != 3 or len(internal) != 625:
            raise RuntimeError("unexpected random.getstate() layout")
        self._bg.state = {"bit_generator": "MT19937", "state": {"key": np.array(internal[:624], dtype=np.uint32), "pos": int(internal[624])}}

    def _store_py(self):
        st = self._bg.state["state"]
        random.setstate((3, tuple(int(v) for v in st["key"]) + (int(st["pos"]),), self.py_state[2]))

    def commit(self, used_py, used_words):
        if used_py != self.n_py:
            self._load_py(self.py_state)
            if used_py:
                self._bg.random_raw(2 * used_py)
        self._store_py()
        if used_words != self.n_words:
            np.random.set_state(self.np_state)
            if used_words:
                np.random.get_bit_generator().random_raw(used_words)

    def rollback(self):
        random.setstate(self.py_state)
        np.random.set_state(self.np_state)


def _car_columns(planner, run_type):
    """(low, high) of the np.random.uniform calls of one non-goal car sample, in call order (base_planner.py:181-191)."""
    W, L, mv = planner.map_width, planner.map_length, planner.max_v
    cols = [(-np.pi, np.pi), (-mv, mv), (-1, 1), (-0.40, 0.40)]
    if run_type < 2:
        cols = [(-W / 2, W / 2), (-L / 2, L / 2)] + cols
    return cols


def _uniform(lo, hi, d):
    return lo + (hi - lo) * d                      # random_uniform(): lower + range * next_double


def draw_round_bulk(planner, B, remain_init_path=None):
    """-> (samples (B, S), cond_goals (B, 2)) or None when this configuration is not covered (the caller then runs the
    reference's loop).  ``planner``: a BasePlanner facade (goal_sample_rate, goal_conditioning_bias, run_type, env, ...)."""
    if B <= 0:
        return None
    ant = "ant" in planner.env_id.lower()
    run_type = int(getattr(planner, "run_type", 0))
    if ant and (run_type != 0 or remain_init_path is not None):
        return None
    S = planner.start_node.state.shape[0]
    goal_state = np.asarray(planner.goal_state, dtype=np.float64)
    gsr, gcb = planner.goal_sample_rate, planner.goal_conditioning_bias
    s = np.zeros((B, S))
    c = np.zeros((B, 2))
    if ant:
        sg, W, L = planner.s_global, planner.map_width, planner.map_length
        cols = [(-sg * W / 2, sg * W / 2), (-sg * L / 2, sg * L / 2)]              # base_planner.py:194-199
    else:
        cols = _car_columns(planner, run_type)
    nd = len(cols) + (1 if (run_type >= 2 and not ant) else 0)                     # doubles per non-goal sample
    cdf = None
    if run_type >= 2 and not ant:
        p = np.array(planner.env.prob_map, dtype=np.float64).ravel()
        cdf = p.cumsum()
        cdf /= cdf[-1]

    def fill_nongoal(rows, d):
        """rows: candidate indices of the non-goal samples, d (len(rows), nd) their doubles in draw order."""
        k = 0
        if cdf is not None:
            idx = cdf.searchsorted(d[:, 0], side="right")
            rr, cc = np.unravel_index(idx, planner.env.prob_map.shape)
            xy = planner.env.cell_rowcol_to_xy(np.array([rr, cc]))
            s[rows, 0], s[rows, 1] = xy[0], xy[1]
            k = 1
            first = 2
        else:
            first = 0
        for q, (lo, hi) in enumerate(cols):
            s[rows, first + q] = _uniform(lo, hi, d[:, k + q])

    if remain_init_path is None:
        # python stream: the goal-sample coin, and (run_type 0) the goal-conditioning coin, per candidate
        per_py = 2 if run_type == 0 else 1
        st = _Streams(per_py * B, 2 * nd * B)
        py = st.py.reshape(B, per_py)
        nongoal = py[:, 0] > gsr
        rows = np.nonzero(nongoal)[0]
        n = rows.size
        d = _doubles(st.words[: 2 * nd * n].reshape(n, nd, 2))
        s[~nongoal] = goal_state                                                   # base_planner.py:201-207
        fill_nongoal(rows, d)
        if run_type == 0:
            own = py[:, 1] > gcb                                                   # RRT.py:153-154
            c[:] = np.where(own[:, None], s[:, :2], goal_state[None, :2])
        else:
            c[:] = s[:, :2]
        st.commit(per_py * B, 2 * nd * n)
        return s, c

    # run_type >= 1 with a remaining reference path (RRT.py:134-137): per candidate np.random.choice of a path point, the
    # explore coin, and -- when exploring -- a random_node_sample.  The consumption of both streams depends on the values, so the
    # positions are resolved by one pass over plain Python lists (no numpy call per candidate), the arithmetic stays vectorised.
    path = np.asarray(remain_init_path)
    n_path = len(path)
    if n_path < 1:
        return None
    rng = n_path - 1
    mask = 0
    while mask < rng:
        mask = (mask << 1) | 1
    n_words = (2 * nd + 6) * B + 4096          # choice: < 2 words on average (rejection rate < 1/2); a miss falls back to the loop
    st = _Streams(2 * B, n_words)
    words = st.words.tolist()
    py = st.py.tolist()
    kp = kw = 0                                                                    # stream positions
    pick = [0] * B
    explore = [False] * B
    is_goal = [False] * B
    dpos = [0] * B                                                                 # word offset of the explore sample's doubles
    try:
        for i in range(B):
            if rng == 0:
                v = 0                                                              # randint(0, 1): no word is drawn
            else:
                while True:
                    v = words[kw] & mask
                    kw += 1
                    if v <= rng:
                        break
            pick[i] = v
            e = py[kp] < 0.4
            kp += 1
            explore[i] = e
            if e:
                g = not (py[kp] > gsr)
                kp += 1
                is_goal[i] = g
                if not g:
                    dpos[i] = kw
                    kw += 2 * nd
                    if kw > n_words:
                        raise IndexError
    except IndexError:                                                             # rejection ran past the block (never in practice)
        st.rollback()
        return None
    explore = np.array(explore)
    is_goal = np.array(is_goal)
    keep = ~explore
    s[keep, :2] = path[np.array(pick)[keep], :2]                                    # remain_init_path[node_idx][np.newaxis]
    s[explore & is_goal] = goal_state
    rows = np.nonzero(explore & ~is_goal)[0]
    if rows.size:
        off = np.array(dpos)[rows]
        w = st.words[(off[:, None] + np.arange(2 * nd)[None, :])].reshape(rows.size, nd, 2)
        fill_nongoal(rows, _doubles(w))
    c[:] = s[:, :2]
    st.commit(kp, kw)
    return s, c
