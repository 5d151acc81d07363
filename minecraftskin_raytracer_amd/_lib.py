"""Loader for the product library libmcrt.so (HIP kernels + C ABI).  No fallback of any kind: if the
library is missing, loading fails loudly."""
from __future__ import annotations

import ctypes as C
import os

from . import abi

PKG = os.path.dirname(os.path.abspath(__file__))
# MCRT_LIB selects another build of the SAME library (kernel tuning experiments); there is still no fallback
LIB_PATH = os.environ.get("MCRT_LIB") or os.path.join(PKG, "libmcrt.so")

_lib = None

# every symbol include/mcrt.h declares
EXPORTED_SYMBOLS = [
    "mcrt_config_init", "mcrt_generate_tiles", "mcrt_abi_version", "mcrt_device_count", "mcrt_last_error",
    "mcrt_render", "mcrt_render_tile", "mcrt_scene_create", "mcrt_scene_destroy", "mcrt_render_device", "mcrt_owned_pixel_rows",
    "mcrt_unpack_rows_device", "mcrt_quantize_rgba8_device", "mcrt_quantize_rgba8", "mcrt_last_timings",
    "mcrt_time_render_device", "mcrt_build_skin_scene", "mcrt_build_default_scene", "mcrt_builtin_pose",
    "mcrt_scene_desc_free", "mcrt_scene_flatten", "mcrt_probe_intersect", "mcrt_probe_trace",
    "mcrt_probe_mt_uniform", "mcrt_probe_detmath", "mcrt_probe_detmath_range", "mcrt_probe_div_const",
    "mcrt_render_device_ex", "mcrt_write_png_rgba8", "mcrt_encode_png_rgba8", "mcrt_write_png_f32", "mcrt_render_png",
    "mcrt_assemble_frame_device", "mcrt_scene_set_lanes", "mcrt_trim", "mcrt_scene_check", "mcrt_render_multi",
    "mcrt_render_rgba8", "mcrt_render_rect",
]


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m minecraftskin_raytracer_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    desc_p = C.POINTER(abi.McrtSceneDesc)
    cfg_p = C.POINTER(abi.McrtConfig)
    u8_p = C.POINTER(C.c_uint8)
    f_p = abi.c_float_p
    vp = C.c_void_p
    sig = {
        "mcrt_config_init": (None, [cfg_p]),
        "mcrt_generate_tiles": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(abi.McrtTile), C.c_int]),
        "mcrt_abi_version": (C.c_int, []),
        "mcrt_device_count": (C.c_int, []),
        "mcrt_last_error": (C.c_char_p, []),
        "mcrt_render": (C.c_int, [desc_p, cfg_p, f_p, abi.PROGRESS_FN, vp, C.c_int]),
        "mcrt_render_tile": (C.c_int, [desc_p, cfg_p, C.c_int, f_p, C.c_int]),
        "mcrt_render_rect": (C.c_int, [desc_p, cfg_p, C.POINTER(abi.McrtTile), f_p, C.c_int]),
        "mcrt_render_rgba8": (C.c_int, [desc_p, cfg_p, u8_p, abi.PROGRESS_FN, vp, C.POINTER(C.c_int), C.c_int, C.c_int]),
        "mcrt_scene_create": (C.c_int, [desc_p, C.c_int, C.POINTER(vp)]),
        "mcrt_scene_destroy": (None, [vp]),
        "mcrt_render_device": (C.c_int, [vp, cfg_p, C.c_int, C.c_int, C.c_int, vp, vp]),
        "mcrt_owned_pixel_rows": (C.c_int, [cfg_p, C.c_int, C.c_int]),
        "mcrt_render_device_ex": (C.c_int, [vp, cfg_p, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
        "mcrt_write_png_rgba8": (C.c_int, [C.c_char_p, u8_p, C.c_int, C.c_int]),
        "mcrt_encode_png_rgba8": (C.c_size_t, [u8_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
        "mcrt_write_png_f32": (C.c_int, [C.c_char_p, f_p, C.c_int, C.c_int]),
        "mcrt_render_png": (C.c_int, [desc_p, cfg_p, C.c_char_p, C.c_int]),
        "mcrt_assemble_frame_device": (C.c_int, [cfg_p, C.c_int, vp, C.c_size_t, vp, vp]),
        "mcrt_scene_set_lanes": (C.c_int, [vp, C.c_int]),
        "mcrt_trim": (None, []),
        "mcrt_scene_check": (C.c_int, [vp]),
        "mcrt_unpack_rows_device": (C.c_int, [cfg_p, C.c_int, C.c_int, vp, vp, vp]),
        "mcrt_quantize_rgba8_device": (C.c_int, [vp, vp, C.c_size_t, vp]),
        "mcrt_quantize_rgba8": (None, [f_p, u8_p, C.c_size_t]),
        "mcrt_last_timings": (C.c_int, [C.POINTER(abi.McrtTimings)]),
        "mcrt_time_render_device": (C.c_int, [vp, cfg_p, C.c_int, C.c_int, C.c_int, vp, vp, C.c_int, f_p]),
        "mcrt_render_multi": (C.c_int, [desc_p, cfg_p, f_p, abi.PROGRESS_FN, vp, C.POINTER(C.c_int), C.c_int, C.c_int]),
        "mcrt_build_skin_scene": (C.c_int, [u8_p, C.c_int, C.c_int, f_p, C.POINTER(desc_p)]),
        "mcrt_build_default_scene": (C.c_int, [f_p, C.POINTER(desc_p)]),
        "mcrt_builtin_pose": (C.c_int, [C.c_int, f_p]),
        "mcrt_scene_desc_free": (None, [desc_p]),
        "mcrt_scene_flatten": (C.c_size_t, [desc_p, vp, C.c_size_t]),
        "mcrt_probe_intersect": (C.c_int, [vp, f_p, C.c_int, vp]),
        "mcrt_probe_trace": (C.c_int, [vp, cfg_p, f_p, C.c_int, C.c_int, f_p]),
        "mcrt_probe_mt_uniform": (C.c_int, [C.c_int, C.POINTER(C.c_uint32), C.c_int, C.c_int, f_p]),
        "mcrt_probe_detmath": (C.c_int, [C.c_int, C.c_int, f_p, f_p, C.c_size_t, f_p]),
        "mcrt_probe_detmath_range": (C.c_int, [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_uint64)]),
        "mcrt_probe_div_const": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class McrtError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"mcrt error {code}: {msg}")
        self.code = code


def check(rc: int) -> None:
    if rc != 0:
        raise McrtError(rc, load().mcrt_last_error().decode("utf-8", "replace"))
