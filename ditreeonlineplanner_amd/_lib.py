"""ctypes binding of libditree_hip.so (include/ditree.h).

The HIP library is the product path: there is no CPU fallback.  Importing this module
never touches the GPU; ``lib()`` raises ``DitreeLibraryError`` when the shared object is
missing or does not export every symbol of the header.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libditree_hip.so")

ST_NOT_RUN, ST_OK, ST_GOAL, ST_COLLIDED = -1, 0, 1, 2
ST_FLAG_GOAL_AT_COLLISION = 0x100
PREC_BF16, PREC_F32, PREC_F16X3, PREC_BF16X3, PREC_F16 = 0, 1, 2, 3, 4
PREC_NAMES = {"bf16": PREC_BF16, "f32": PREC_F32, "f16x3": PREC_F16X3, "bf16x3": PREC_BF16X3, "f16": PREC_F16}


class DitreeLibraryError(RuntimeError):
    pass


class DitreeError(RuntimeError):
    pass


class Tree(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n_chunks", C.c_int32), ("A", C.c_int32),
                ("state_dim", C.c_int32), ("action_dim", C.c_int32),
                ("state", C.c_void_p), ("xy", C.c_void_p), ("parent", C.c_void_p),
                ("last_action", C.c_void_p), ("has_prev", C.c_void_p), ("num_visit", C.c_void_p),
                ("edge_states", C.c_void_p), ("edge_actions", C.c_void_p),
                ("edge_nstates", C.c_void_p), ("edge_nactions", C.c_void_p), ("obstacle_ahead", C.c_void_p),
                ("edge_owner", C.c_void_p), ("hist", C.c_void_p), ("hist_n", C.c_void_p), ("counters", C.c_void_p)]


class Round(C.Structure):
    _fields_ = [("B", C.c_int32), ("parent", C.c_void_p), ("status", C.c_void_p),
                ("chunks_run", C.c_void_p), ("end_state", C.c_void_p), ("states", C.c_void_p),
                ("actions", C.c_void_p), ("chunk_steps", C.c_void_p), ("node_id", C.c_void_p),
                ("last_action", C.c_void_p), ("first_action", C.c_void_p),
                ("own_lo", C.c_int32), ("own_n", C.c_int32), ("shard", C.c_int32),
                ("hist", C.c_void_p), ("hist_n", C.c_void_p)]


RECORD_DOUBLES = 12          # include/ditree.h DITREE_RECORD_DOUBLES (the car's record; ditree_record_doubles for any tree)


class Strides(C.Structure):
    """include/ditree.h ditree_strides: element (candidate b, row i, component k) at b * cand + i * row + k * comp doubles."""
    _fields_ = [("cand", C.c_int64), ("row", C.c_int64), ("comp", C.c_int64)]


class AntModel(C.Structure):
    """include/ditree.h ditree_ant_model -- the build's stand-in for the ant's MuJoCo step (NOT MuJoCo, parity unpinned)."""
    _fields_ = [(n, C.c_double) for n in ("h", "frame_skip", "k_act", "k_spr", "k_dmp", "k_lim", "hip_lim", "ank_lo", "ank_hi",
                                          "ank_rest", "contact_gain", "leg_r", "k_push", "c_lin", "z0", "z_gain", "k_z", "c_z",
                                          "k_lift", "c_ang", "k_up", "k_yaw", "cphi", "sphi")]

    @classmethod
    def default(cls):
        return cls(0.01, 5.0, 60.0, 20.0, 6.0, 400.0, 0.5236, 0.5236, 1.2217, 0.87, 6.0, 0.4, 16.0, 3.0, 0.55, 0.3, 120.0, 12.0,
                   4.0, 5.0, 25.0, 10.0, 0.5 ** 0.5, 0.5 ** 0.5)


ANT_DYN_TAPE, ANT_DYN_MODEL = 0, 1


class RoundParams(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("samples", C.c_void_p), ("cond_goal", C.c_void_p),
                ("noise", C.c_void_p), ("inject_actions", C.c_void_p), ("P", C.c_int32), ("K", C.c_int32),
                ("t0", C.POINTER(C.c_float)), ("dt", C.POINTER(C.c_float)), ("norm", C.POINTER(C.c_double)),
                ("goal_xy", C.POINTER(C.c_double)), ("axis", C.POINTER(C.c_double)), ("lm_n", C.c_int32),
                ("lm_size", C.c_double), ("s_global", C.c_double), ("early_exit", C.c_int32),
                ("ddpm_coef", C.POINTER(C.c_float)), ("step_noise", C.c_void_p), ("chunk_budget", C.c_void_p)]


class AntRoundParams(C.Structure):
    _fields_ = [("n_nodes", C.c_int32), ("samples", C.c_void_p), ("cond_goal", C.c_void_p), ("noise", C.c_void_p),
                ("inject_actions", C.c_void_p), ("P", C.c_int32), ("K", C.c_int32), ("t0", C.POINTER(C.c_float)),
                ("dt", C.POINTER(C.c_float)), ("norm", C.POINTER(C.c_double)), ("desired_goal", C.POINTER(C.c_double)),
                ("goal_radius", C.c_double), ("ball_radius", C.c_double), ("axis", C.POINTER(C.c_double)), ("lm_n", C.c_int32),
                ("lm_size", C.c_double), ("s_global", C.c_double), ("dynamics", C.c_int32), ("next_obs_tape", C.c_void_p),
                ("model", C.POINTER(AntModel)), ("early_exit", C.c_int32), ("ddpm_coef", C.POINTER(C.c_float)),
                ("step_noise", C.c_void_p), ("cond_out", C.c_void_p)]


class MppiParams(C.Structure):
    _fields_ = [("T", C.c_int32), ("K", C.c_int32), ("lam", C.c_double), ("sigma", C.c_double * 2),
                ("w_track", C.c_double), ("w_progress", C.c_double), ("w_collision", C.c_double), ("w_goal", C.c_double),
                ("seed", C.c_uint64), ("window_back", C.c_int32), ("window_fwd", C.c_int32), ("lanes", C.c_int32), ("k_offset", C.c_int64)]


class MppiAntParams(C.Structure):
    _fields_ = [("T", C.c_int32), ("K", C.c_int32), ("lam", C.c_double), ("sigma", C.c_double * 8), ("w_track", C.c_double),
                ("w_progress", C.c_double), ("w_collision", C.c_double), ("w_goal", C.c_double), ("seed", C.c_uint64),
                ("window_back", C.c_int32), ("window_fwd", C.c_int32), ("k_offset", C.c_int64), ("goal_radius", C.c_double),
                ("ball_radius", C.c_double), ("s_global", C.c_double), ("model", AntModel)]


MPPI_ROLLOUTS, MPPI_MIN, MPPI_SUMS, MPPI_APPLY, MPPI_EXECUTE, MPPI_ALL = 1, 2, 4, 8, 16, 31


_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pd, _pf = C.POINTER(C.c_double), C.POINTER(C.c_float)

# name -> (restype, argtypes); must list every symbol include/ditree.h declares
SIGNATURES = {
    "ditree_version": (_i32, []),
    "ditree_build_id": (C.c_char_p, []),
    "ditree_ctx_create": (_i32, [_i32, C.POINTER(_vp)]),
    "ditree_ctx_destroy": (None, [_vp]),
    "ditree_last_error": (C.c_char_p, [_vp]),
    "ditree_upload_maze": (_i32, [_vp, _pf, _i32, _i32, _vp]),
    "ditree_nn_argmin": (_i32, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ditree_local_map": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _pd, _f64, _i32, _vp, _vp]),
    "ditree_chunk_budget": (_i32, [_vp, C.POINTER(Tree), _vp, _i32, _i32, C.POINTER(_i32), _i32, _vp, _vp, _vp]),
    "ditree_cond_vector_ant": (_i32, [_vp, _vp, _i32, _vp, _vp, _vp, _i32, _pd, _f64, _vp, _vp]),
    "ditree_cond_vector": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _pd, _f64, _vp, _vp]),
    "ditree_car_rollout": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _pd, _vp, _i64, _vp, _i64, _vp, _vp,
                                  _vp, _vp]),
    "ditree_car_rollout_ld": (_i32, [_vp, _vp, _vp, _i64, _vp, _i32, _i32, _pd, _vp, C.POINTER(Strides), _vp, C.POINTER(Strides),
                                     _vp, _vp, _vp, _vp]),
    "ditree_ant_collision": (_i32, [_vp, _vp, _i32, _i32, _f64, _f64, _vp, _vp]),
    "ditree_ant_rollout": (_i32, [_vp, C.POINTER(AntModel), _vp, _vp, _i64, _vp, _i64, _vp, _i32, _i32, _pd, _f64, _f64, _f64,
                                  _vp, C.POINTER(Strides), _vp, C.POINTER(Strides), _vp, _vp]),
    "ditree_lidar_scan": (_i32, [_vp, _vp, _i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "ditree_obstacle_ahead": (_i32, [_vp, _vp, _i32, _i32, _vp, _vp]),
    "ditree_path_after_obstacle": (_i32, [_vp, _vp, _i32, _i32, _pd, _i32, _vp, _vp]),
    "ditree_fallback_select": (_i32, [_vp, C.POINTER(Tree), _i32, _pd, _pd, _i32, _vp, _vp]),
    "ditree_follow_plan": (_i32, [_vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _pd, _f64, _f64, _vp, _vp, _vp]),
    "ditree_accept": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), _i32, _vp]),
    "ditree_round_pack": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), _vp, _vp]),
    "ditree_round_unpack": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), _vp, _vp]),
    "ditree_record_doubles": (_i32, [C.POINTER(Tree)]),
    "ditree_comm_unique_id": (_i32, [_vp, _vp]),
    "ditree_comm_init": (_i32, [_vp, _i32, _i32, _vp]),
    "ditree_allgather_nodes": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "ditree_comm_destroy": (_i32, [_vp]),
    "ditree_mppi_step": (_i32, [_vp, C.POINTER(MppiParams), _vp, _vp, _vp, _i32, _pd, _vp, C.c_uint64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ditree_mppi_step_ant": (_i32, [_vp, C.POINTER(MppiAntParams), _vp, _vp, _vp, _i32, _pd, _vp, C.c_uint64, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ditree_load_weights": (_i32, [_vp, _vp, _i64, C.c_char_p, _vp]),
    "ditree_denoise_reserve": (_i32, [_vp, _i32, _i32]),
    "ditree_denoise": (_i32, [_vp, _vp, _vp, _vp, _i32, _i32, _pf, _pf, _pd, _vp, _vp, _vp]),
    "ditree_denoise_ddpm": (_i32, [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _pf, _pf, _pd, _vp, _vp, _vp]),
    "ditree_denoise_eval": (_i32, [_vp, _vp, _vp, _vp, _i32, C.c_float, _i32, _vp, _vp]),
    "ditree_denoise_dims": (_i32, [_vp, C.POINTER(_i32)]),
    "ditree_denoise_status": (_i32, [_vp, C.POINTER(_i32), C.c_char_p, _i64, _i32, _vp]),
    "ditree_profile": (_i32, [_vp, _i32]),
    "ditree_profile_read": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "ditree_denoise_debug_read": (_i32, [_vp, C.c_char_p, _i32, _vp, _i64, C.POINTER(_i32), _vp]),
    "ditree_expand_round_ant": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), C.POINTER(AntRoundParams), _vp]),
    "ditree_ant_round_begin": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), C.POINTER(AntRoundParams), _vp]),
    "ditree_ant_chunk_sample": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), C.POINTER(AntRoundParams), _i32, _vp]),
    "ditree_ant_chunk_step": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), C.POINTER(AntRoundParams), _i32, _vp, _vp]),
    "ditree_round_stats": (_i32, [_vp, C.POINTER(_i32)]),
    "ditree_expand_round": (_i32, [_vp, C.POINTER(Tree), C.POINTER(Round), C.POINTER(RoundParams), _vp]),
}

_LIB = None


def lib():
    """Load (once) and return the ctypes handle; fail loudly if it is not there."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime of the process (tensors and
    # streams come from torch), so make sure it is mapped before our library resolves the same SONAME.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise DitreeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -m ditreeonlineplanner_amd.build` "
            "(hipcc, gfx950).  There is no CPU fallback for the expansion path.")
    # a library built from other sources than the ones next to it (a stale .so that travelled with a snapshot) is refused
    # (DITREE_ALLOW_STALE_LIB=1 skips the check; a deployment that ships the package and the .so WITHOUT csrc/ and
    # include/ditree.h has nothing to compare against: the check is skipped with a warning, the library's own id is what
    # `build_id()` reports)
    if os.environ.get("DITREE_ALLOW_STALE_LIB", "0") != "1":
        from . import build as _build
        try:
            want = _build.source_id()
        except OSError as e:
            import warnings
            warnings.warn(f"libditree_hip.so: sources not found next to the package ({e.filename}); stale-library check skipped")
            want = None
        have = _build.library_id(LIB_PATH)
        if want is not None and have != want:
            raise DitreeLibraryError(f"{LIB_PATH} is stale: built from sources {have}, the sources here are {want}; "
                                     "run `python -m ditreeonlineplanner_amd.build`")
    try:
        h = C.CDLL(LIB_PATH)
    except OSError as e:
        raise DitreeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(h, name)
        except AttributeError as e:
            raise DitreeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _LIB = h
    return h


def build_id() -> str:
    return lib().ditree_build_id().decode()


def check(ctx_handle, rc: int, what: str = ""):
    if rc != 0:
        msg = lib().ditree_last_error(ctx_handle)
        raise DitreeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
