"""ctypes binding of libmpdx.so (C ABI declared in include/mpdx.h).

The product path has NO fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import os as _os

# MPDX_LIB=<path> selects another build of the library (A/B comparisons of kernel variants)
_LIB_PATH = Path(_os.environ["MPDX_LIB"]) if _os.environ.get("MPDX_LIB") else Path(__file__).resolve().parent / "libmpdx.so"
_lib = None

MAX_LEVELS = 8


class UnetCfg(C.Structure):
    _fields_ = [("state_dim", C.c_int32), ("n_support_points", C.c_int32), ("unet_input_dim", C.c_int32),
                ("n_levels", C.c_int32), ("dim_mults", C.c_int32 * MAX_LEVELS), ("time_emb_dim", C.c_int32)]


class StepCoefs(C.Structure):
    _fields_ = [("sqrt_recip_alphas_cumprod", C.c_float), ("sqrt_recipm1_alphas_cumprod", C.c_float),
                ("posterior_mean_coef1", C.c_float), ("posterior_mean_coef2", C.c_float),
                ("noise_scale", C.c_float), ("noise_std_extra", C.c_float),
                ("predict_epsilon", C.c_int32), ("clip_denoised", C.c_int32), ("ddim_k1", C.c_float), ("ddim_k2", C.c_float),
                ("guide_scale", C.c_float)]


MAX_FIELDS = 4
FIELD_OBJECTS, FIELD_WORKSPACE, FIELD_SELF = 0, 1, 2
ROBOT_POINTMASS, ROBOT_PANDA = 0, 1


class Field(C.Structure):
    _fields_ = [("kind", C.c_int32), ("weight", C.c_float), ("sphere_off", C.c_int32), ("n_spheres", C.c_int32),
                ("box_off", C.c_int32), ("n_boxes", C.c_int32), ("ws_min", C.c_float * 3), ("ws_max", C.c_float * 3)]


class GuideParams(C.Structure):
    _fields_ = [("robot", C.c_int32), ("q_dim", C.c_int32), ("ws_dim", C.c_int32), ("interpolate", C.c_int32),
                ("n_interp", C.c_int32), ("clip_grad", C.c_int32), ("max_grad_norm", C.c_float),
                ("mins", C.c_float * 16), ("maxs", C.c_float * 16), ("cutoff_margin", C.c_float), ("link_margin", C.c_float),
                ("n_fields", C.c_int32), ("fields", Field * MAX_FIELDS), ("use_gp", C.c_int32), ("gp_weight", C.c_float),
                ("dt", C.c_float), ("sigma_gp", C.c_float), ("prims", C.c_void_p), ("n_prim_floats", C.c_int32),
                ("clip_rule", C.c_int32), ("max_grad_value", C.c_float), ("gp_half_factor", C.c_int32),
                ("identity_normalizer", C.c_int32)]


class GpmpOpts(C.Structure):
    _fields_ = [("sigma_obs", C.c_float), ("lambda_up", C.c_float), ("lambda_down", C.c_float), ("lambda_min", C.c_float),
                ("lambda_max", C.c_float), ("step", C.c_float), ("adaptive", C.c_int32)]


class RrtOpts(C.Structure):
    _fields_ = [("q_lo", C.c_float * 8), ("q_hi", C.c_float * 8), ("step", C.c_float), ("max_nodes", C.c_int32), ("max_iters", C.c_int32),
                ("max_connect_steps", C.c_int32), ("n_edge_checks", C.c_int32), ("seed", C.c_uint64)]


# every symbol include/mpdx.h declares: name -> (restype, argtypes)
_vp, _sz, _i, _f = C.c_void_p, C.c_size_t, C.c_int, C.c_float
SIGNATURES = {
    "mpdx_last_error": (C.c_char_p, []),
    "mpdx_version": (_i, []),
    "mpdx_unet_create": (_i, [C.POINTER(UnetCfg), C.POINTER(_vp)]),
    "mpdx_unet_destroy": (None, [_vp]),
    "mpdx_unet_num_params": (_i, [_vp]),
    "mpdx_unet_param_info": (_i, [_vp, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int32 * 3), C.POINTER(C.c_int32)]),
    "mpdx_unet_packed_floats": (_sz, [_vp]),
    "mpdx_unet_timetab_floats": (_sz, [_vp, _i]),
    "mpdx_unet_workspace_floats": (_sz, [_vp, _i]),
    "mpdx_unet_pack_param": (_i, [_vp, C.c_char_p, _vp, _sz, _vp, _vp]),
    "mpdx_unet_build_timetab": (_i, [_vp, _vp, _vp, _i, _vp, _vp]),
    "mpdx_unet_forward": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp]),
    "mpdx_ddpm_step": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, C.POINTER(StepCoefs), _i, _i, _vp, _vp, _i, _i, _vp, _vp]),
    "mpdx_add_noise": (_i, [_vp, _vp, _vp, _vp, _f, _f, _vp, _i, _i, _i, _vp]),
    "mpdx_hard_conds": (_i, [_vp, _vp, _i, C.POINTER(C.c_int32), C.POINTER(_vp), _i, _i, _i, _vp]),
    "mpdx_q_sample": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_weighted_loss": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "mpdx_plan": (_i, [_vp, _vp, _vp, _i, C.POINTER(StepCoefs), _i, _vp, _vp, _vp, _vp, _vp, _i, _vp,
                        C.POINTER(GuideParams), _i, _i, _vp, _i, C.c_uint64, C.c_uint64, _vp]),
    "mpdx_guide_step": (_i, [C.POINTER(GuideParams), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_guide_step_scaled": (_i, [C.POINTER(GuideParams), _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp]),
    "mpdx_guide_time": (_i, [C.POINTER(GuideParams), _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, C.POINTER(C.c_float)]),
    "mpdx_traj_metrics": (_i, [C.POINTER(GuideParams), _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_traj_metrics_mask": (_i, [C.POINTER(GuideParams), _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_guide_trace": (_i, [C.POINTER(GuideParams), _vp, _vp, _i, _i, _i, _vp, C.POINTER(C.c_longlong)]),
    "mpdx_absmax": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_unet_profile": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, C.POINTER(C.c_float), C.POINTER(C.c_double),
                                C.POINTER(C.c_char_p), C.POINTER(C.c_int)]),
    "mpdx_layer_trace": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(C.c_longlong)]),
    "mpdx_fused_trace": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, C.POINTER(C.c_longlong), _i, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mpdx_unet_time_units": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, _i, _i, _i, C.POINTER(C.c_float)]),
    "mpdx_unet_time_without": (_i, [_vp, _vp, _vp, _i, _vp, _i, _i, _vp, _vp, C.c_uint64, _i, C.POINTER(C.c_float)]),
    "mpdx_unet_unit_layer": (_i, [_vp, _i, _i]),
    "mpdx_unet_unit_is_pair": (_i, [_vp, _i, _i]),
    "mpdx_unet_unit_bytes": (C.c_double, [_vp, _i, _i]),
    "mpdx_unet_fused_program": (_i, [_vp, _i]),
    "mpdx_bench_layer": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _i, _i, C.POINTER(C.c_float)]),
    "mpdx_unet_layer_tile": (_i, [_vp, _i, _i, C.c_char_p, _sz]),
    "mpdx_randn": (_i, [_vp, _sz, C.c_uint64, C.c_uint64, _vp]),
    "mpdx_train_flat_floats": (_sz, [_vp]),
    "mpdx_train_dgrad_pack_floats": (_sz, [_vp]),
    "mpdx_train_workspace_floats": (_sz, [_vp, _i]),
    "mpdx_train_param_offset": (_i, [_vp, _i, C.POINTER(_sz), C.POINTER(_sz)]),
    "mpdx_train_pack": (_i, [_vp, _vp, _vp, _vp, _vp]),
    "mpdx_train_loss_backward": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp]),
    "mpdx_adam_step": (_i, [_vp, _vp, _vp, _vp, _sz, _f, _f, _f, _f, _i, _f, _vp, _vp]),
    "mpdx_train_draw": (_i, [_vp, C.c_uint64, _vp]),
    "mpdx_ema_update": (_i, [_vp, _vp, _sz, _f, _vp]),
    "mpdx_gpmp_step": (_i, [C.POINTER(GuideParams), C.POINTER(GpmpOpts), _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mpdx_rrt_paths": (_i, [C.POINTER(GuideParams), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _i, _i, _vp]),
    "mpdx_rrt_connect": (_i, [C.POINTER(GuideParams), C.POINTER(RrtOpts), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp]),
}


def lib_path() -> Path:
    return _LIB_PATH


class LibraryUnavailable(RuntimeError):
    """libmpdx.so is missing or cannot be loaded on this host (every compute entry point raises it; nothing falls back)"""


def load():
    """Load libmpdx.so and bind every declared symbol; raises LibraryUnavailable (a RuntimeError; never falls back) when unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise LibraryUnavailable(f"{_LIB_PATH} is missing - build it with `python -m mpd_public_amd.build` (hipcc, gfx950). "
                           "mpd_public_amd has no CPU fallback.")
    try:
        lib = C.CDLL(str(_LIB_PATH))
    except OSError as e:  # pragma: no cover
        raise LibraryUnavailable(f"cannot load {_LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise LibraryUnavailable(f"{_LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().mpdx_last_error()
        raise RuntimeError(f"libmpdx {what} failed (code {rc}): {msg.decode() if msg else '?'}")


def ptr(t) -> int:
    """Raw device pointer of a contiguous fp32 CUDA(HIP) tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
