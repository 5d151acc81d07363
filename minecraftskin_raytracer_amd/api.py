"""Host-side mirror of the reference's interface for the render hot path.

Names follow the reference (``TileRenderer.render/generateTiles/lastErrors``, ``Config`` =
``RayTracer::Config``, ``MeshBuilder.buildScene/buildDefaultScene``), argument meaning and error
behaviour too; every render goes through the C ABI of ``libmcrt.so`` into the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import abi
from ._lib import McrtError, check, load
from .abi import Config, Mesh, Scene, SceneDescHolder, Texture

Image = np.ndarray  # (H, W, 4) float32 — skin/image.h:9-15


def trim() -> None:
    """Free the device workspace kept for reuse by destroyed scenes (mcrt_trim)."""
    load().mcrt_trim()


def device_count() -> int:
    return int(load().mcrt_device_count())


class _OwnedDescPtr(C.POINTER(abi.McrtSceneDesc)):
    """A ``POINTER(McrtSceneDesc)`` that keeps the object owning the pointed-to arrays alive: a ``.ptr`` taken
    from a temporary (``skin_scene(...).ptr``) stays valid for as long as the pointer itself is referenced."""

    _type_ = abi.McrtSceneDesc


class SceneDesc:
    """A scene description the library can consume: either built from a Python ``Scene`` or owned by
    the native scene builder.  ``.ptr`` is a ``POINTER(McrtSceneDesc)``-compatible object that holds a
    reference to this description."""

    def __init__(self, scene: Optional[Scene] = None, _native=None):
        self._holder = SceneDescHolder(scene) if scene is not None else None
        self._native = _native

    @property
    def ptr(self):
        raw = self._native if self._native is not None else C.pointer(self._holder.desc)
        p = C.cast(raw, _OwnedDescPtr)
        p._owner = self  # the description (and through it the arrays) lives as long as the pointer
        return p

    @property
    def desc(self) -> abi.McrtSceneDesc:
        return self._native.contents if self._native is not None else self._holder.desc

    def to_numpy(self) -> dict:
        return abi.desc_to_numpy(self.desc)

    def __del__(self):
        if getattr(self, "_native", None) is not None:
            try:
                load().mcrt_scene_desc_free(self._native)
            except Exception:
                pass
            self._native = None


def _as_desc(scene) -> SceneDesc:
    if isinstance(scene, SceneDesc):
        return scene
    if isinstance(scene, Scene):
        return SceneDesc(scene)
    raise TypeError("scene must be a Scene or SceneDesc")


class MeshBuilder:
    """scene/mesh_builder.h:7-34 (scene construction from skin data; native implementation)."""

    @staticmethod
    def buildScene(skin_rgba8: np.ndarray, pose: Optional[Sequence[float]] = None) -> SceneDesc:
        """SkinParser::parse (64x64 / 64x32 RGBA8) + MeshBuilder::buildScene."""
        skin = np.ascontiguousarray(skin_rgba8, np.uint8)
        if skin.ndim != 3 or skin.shape[2] != 4:
            raise ValueError("skin must be (H, W, 4) uint8")
        h, w = skin.shape[:2]
        p = np.asarray(pose if pose is not None else [0.0] * 12, np.float32)
        out = C.POINTER(abi.McrtSceneDesc)()
        rc = load().mcrt_build_skin_scene(skin.ctypes.data_as(C.POINTER(C.c_uint8)), w, h, abi.fptr(p), C.byref(out))
        if rc != 0:
            # skin_parser.cpp:122-131: only 64x64 and 64x32 are valid
            raise ValueError(f"Invalid skin dimensions: {w}x{h} (expected 64x64 or 64x32)")
        return SceneDesc(_native=out)

    @staticmethod
    def buildDefaultScene(pose: Optional[Sequence[float]] = None) -> SceneDesc:
        p = np.asarray(pose if pose is not None else [0.0] * 12, np.float32)
        out = C.POINTER(abi.McrtSceneDesc)()
        check(load().mcrt_build_default_scene(abi.fptr(p), C.byref(out)))
        return SceneDesc(_native=out)


def getBuiltinPoses() -> List[np.ndarray]:
    """scene/pose.h:25-92 → 7 poses x 12 floats ({head, body, rArm, lArm, rLeg, lLeg} x {rotX, rotZ})."""
    out = []
    for i in range(7):
        p = np.zeros(12, np.float32)
        check(load().mcrt_builtin_pose(i, abi.fptr(p)))
        out.append(p)
    return out


def _devices(device):
    """``device`` argument of the render entry points → (one index, None) or (None, list of indices; [] = every visible
    device).  Accepts any integer type (numpy, torch device indices), ``"all"`` / ``-1``, or a sequence of indices."""
    import operator

    if isinstance(device, str):
        if device != "all":
            raise ValueError("device must be an index, 'all' or a sequence of indices")
        return None, []
    try:
        d = operator.index(device)
    except TypeError:
        devs = [operator.index(x) for x in device]
        if any(x < 0 for x in devs):
            raise ValueError("device indices must be >= 0")
        return None, devs
    if d == -1:
        return None, []
    if d < 0:
        raise ValueError("device must be >= 0, or -1 / 'all' for every visible device")
    return d, None


class TileRenderer:
    """raytracer/tile_renderer.h:16-47."""

    _errors: List[Tuple[int, str]] = []

    @staticmethod
    def generateTiles(imageWidth: int, imageHeight: int, tileSize: int) -> List[Tuple[int, int, int, int]]:
        lib = load()
        n = lib.mcrt_generate_tiles(imageWidth, imageHeight, tileSize, None, 0)
        arr = (abi.McrtTile * max(n, 1))()
        lib.mcrt_generate_tiles(imageWidth, imageHeight, tileSize, arr, n)
        return [(arr[i].x, arr[i].y, arr[i].width, arr[i].height) for i in range(n)]

    @staticmethod
    def render(scene, config: Config, progressCallback: Optional[Callable[[int, int], None]] = None,
               device=0, gather: bool = False, out: Optional[np.ndarray] = None) -> Image:
        """TileRenderer::render.  Per-frame failures never raise (tile_renderer.cpp:158-166): they are
        recorded as one ``(-1, message)`` entry in ``lastErrors()`` and the image keeps Color() =
        (0,0,0,1) pixels.

        ``device``: an index, ``"all"`` / ``-1`` (every visible device) or a sequence of indices — one rank per
        entry, cyclic tile rows (mcrt_render_multi; ``gather`` selects the peer-copy assembly on the first
        device instead of per-device downloads).  ``out``: a (H, W, 4) float32 C-contiguous array to render
        into (the reference returns a fresh Image per call; a caller that renders repeatedly can keep one)."""
        lib = load()
        d = _as_desc(scene)
        c = config.to_c()
        w, h = max(config.width, 0), max(config.height, 0)
        if out is None:
            out = np.zeros((h, w, 4), np.float32)
            out[..., 3] = 1.0  # Image(w,h): default Color() = (0,0,0,1)
        elif out.shape != (h, w, 4) or out.dtype != np.float32 or not out.flags["C_CONTIGUOUS"]:
            raise ValueError("out must be a C-contiguous (height, width, 4) float32 array")
        TileRenderer._errors = []
        if w == 0 or h == 0 or config.tileSize <= 0:
            return out
        cb = abi.PROGRESS_FN((lambda done, total, _u: progressCallback(done, total))) if progressCallback else C.cast(None, abi.PROGRESS_FN)
        one, devs = _devices(device)
        if one is not None:
            rc = lib.mcrt_render(d.ptr, C.byref(c), abi.fptr(out), cb, None, one)
        else:
            arr = (C.c_int * max(len(devs), 1))(*devs)
            rc = lib.mcrt_render_multi(d.ptr, C.byref(c), abi.fptr(out), cb, None, arr if devs else None, len(devs), 1 if gather else 0)
        if rc != 0:
            TileRenderer._errors = [(-1, lib.mcrt_last_error().decode("utf-8", "replace"))]
            out[...] = 0.0
            out[..., 3] = 1.0
        return out

    @staticmethod
    def renderTile(tile: Tuple[int, int, int, int], scene, config: Config, output: np.ndarray, device: int = 0) -> None:
        """TileRenderer::renderTile (tile_renderer.cpp:71-127): one Tile ``(x, y, width, height)`` — any rectangle of the
        frame, its own mt19937(y * width + x) — into ``output`` ((H, W, 4) float32), every other pixel untouched.
        Failures are recorded in ``lastErrors()``."""
        if output.shape != (config.height, config.width, 4) or output.dtype != np.float32 or not output.flags["C_CONTIGUOUS"]:
            raise ValueError("output must be a C-contiguous (height, width, 4) float32 array")
        c = config.to_c()
        t = abi.McrtTile(*[int(v) for v in tile])
        lib = load()
        if lib.mcrt_render_rect(_as_desc(scene).ptr, C.byref(c), C.byref(t), abi.fptr(output), int(device)) != 0:
            TileRenderer._errors.append((-1, lib.mcrt_last_error().decode("utf-8", "replace")))

    @staticmethod
    def renderRGBA8(scene, config: Config, device=0, gather: bool = False) -> np.ndarray:
        """The frame as the RGBA8 plane ImageWriter::writePNG would encode ((H, W, 4) uint8), quantised in the kernels'
        epilogue: 4 B per pixel over PCIe / xGMI instead of 16 (mcrt_render_rgba8).  ``device`` as for ``render``."""
        lib = load()
        c = config.to_c()
        w, h = max(config.width, 0), max(config.height, 0)
        out = np.zeros((h, w, 4), np.uint8)
        out[..., 3] = 255
        TileRenderer._errors = []
        if w == 0 or h == 0 or config.tileSize <= 0:
            return out
        one, devs = _devices(device)
        if one is not None:
            devs = [one]
        arr = (C.c_int * max(len(devs), 1))(*devs)
        rc = lib.mcrt_render_rgba8(_as_desc(scene).ptr, C.byref(c), out.ctypes.data_as(C.POINTER(C.c_uint8)), C.cast(None, abi.PROGRESS_FN), None,
                                   arr if devs else None, len(devs), 1 if gather else 0)
        if rc != 0:
            TileRenderer._errors = [(-1, lib.mcrt_last_error().decode("utf-8", "replace"))]
            out[...] = 0
            out[..., 3] = 255
        return out

    @staticmethod
    def lastErrors() -> List[Tuple[int, str]]:
        return list(TileRenderer._errors)

    @staticmethod
    def lastTimings() -> dict:
        t = abi.McrtTimings()
        load().mcrt_last_timings(C.byref(t))
        return {k: getattr(t, k) for k, _ in abi.McrtTimings._fields_}


def quantize_rgba8(image: np.ndarray) -> np.ndarray:
    """ImageWriter quantiser (image_writer.cpp:18-22): (H, W, 4) float32 → uint8."""
    img = np.ascontiguousarray(image, np.float32)
    out = np.zeros(img.shape, np.uint8)
    load().mcrt_quantize_rgba8(abi.fptr(img), out.ctypes.data_as(C.POINTER(C.c_uint8)), img.size // 4)
    return out


class ImageWriter:
    """The reference's PNG hand-off (src/output/image_writer.h): quantise + write.  The file is an
    uncompressed (zlib-stored) 8-bit RGBA PNG written by the library's own store-only encoder."""

    @staticmethod
    def writePNG(image: np.ndarray, path: str) -> bool:
        """(H, W, 4) float32 → PNG at `path`.  False on failure (empty image, unwritable path), like
        ImageWriter::writePNG (image_writer.cpp:6-28)."""
        img = np.ascontiguousarray(image, np.float32)
        if img.ndim != 3 or img.shape[2] != 4 or img.shape[0] <= 0 or img.shape[1] <= 0:
            return False
        return load().mcrt_write_png_f32(os.fsencode(path), abi.fptr(img), img.shape[1], img.shape[0]) == 0

    @staticmethod
    def writePNG8(rgba8: np.ndarray, path: str) -> bool:
        img = np.ascontiguousarray(rgba8, np.uint8)
        if img.ndim != 3 or img.shape[2] != 4 or img.shape[0] <= 0 or img.shape[1] <= 0:
            return False
        return load().mcrt_write_png_rgba8(os.fsencode(path), img.ctypes.data_as(C.POINTER(C.c_uint8)), img.shape[1], img.shape[0]) == 0

    @staticmethod
    def encodePNG8(rgba8: np.ndarray) -> bytes:
        img = np.ascontiguousarray(rgba8, np.uint8)
        h, w = img.shape[:2]
        p = img.ctypes.data_as(C.POINTER(C.c_uint8))
        n = load().mcrt_encode_png_rgba8(p, w, h, None, 0)
        buf = (C.c_uint8 * n)()
        got = load().mcrt_encode_png_rgba8(p, w, h, buf, n)
        return bytes(buf[:got])


def render_png(scene, config: Config, path: str, device: int = 0) -> bool:
    """TileRenderer::render + ImageWriter::writePNG in one call (RGBA8 quantised in the kernel epilogue,
    4 B/pixel copied back)."""
    c = config.to_c()
    return load().mcrt_render_png(_as_desc(scene).ptr, C.byref(c), os.fsencode(path), device) == 0


class DeviceScene:
    """A flattened scene resident in HBM on one device (mcrt_scene).  Renders go to device pointers
    (e.g. ``torch.Tensor.data_ptr()``) on a caller-chosen HIP stream."""

    def __init__(self, scene, device: int = 0):
        self._desc = _as_desc(scene)
        self._h = C.c_void_p()
        self.device = device
        check(load().mcrt_scene_create(self._desc.ptr, device, C.byref(self._h)))

    def close(self):
        if self._h:
            load().mcrt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self) -> None:
        """Wait for the scene's device work; raises if the device flagged an internal inconsistency."""
        check(load().mcrt_scene_check(self._h))

    def set_lanes(self, lanes: int) -> None:
        """0 = automatic split of a render over internal streams, n >= 1 = exactly n (mcrt_scene_set_lanes)."""
        check(load().mcrt_scene_set_lanes(self._h, int(lanes)))

    def owned_pixel_rows(self, config: Config, first: int = 0, step: int = 1) -> int:
        c = config.to_c()
        return int(load().mcrt_owned_pixel_rows(C.byref(c), first, step))

    def render_device(self, config: Config, out_ptr: int, first: int = 0, step: int = 1,
                      layout: int = abi.LAYOUT_FRAME, stream: int = 0) -> None:
        c = config.to_c()
        check(load().mcrt_render_device(self._h, C.byref(c), first, step, layout, C.c_void_p(out_ptr), C.c_void_p(stream)))

    def render_device_ex(self, config: Config, out_f32_ptr: int = 0, out_rgba8_ptr: int = 0, first: int = 0, step: int = 1,
                         layout: int = abi.LAYOUT_FRAME, stream: int = 0) -> None:
        """Render with the quantisation fused into the epilogue: float4 frame and/or RGBA8 plane."""
        c = config.to_c()
        check(load().mcrt_render_device_ex(self._h, C.byref(c), first, step, layout, C.c_void_p(out_f32_ptr or None),
                                           C.c_void_p(out_rgba8_ptr or None), C.c_void_p(stream)))

    def time_render_device(self, config: Config, out_ptr: int, iters: int, first: int = 0, step: int = 1,
                           layout: int = abi.LAYOUT_FRAME, stream: int = 0) -> float:
        """Average ms of one render's whole pipeline on the device, measured with hipEvents on `stream`."""
        c = config.to_c()
        a = C.c_float()
        check(load().mcrt_time_render_device(self._h, C.byref(c), first, step, layout, C.c_void_p(out_ptr),
                                             C.c_void_p(stream), iters, C.byref(a)))
        return float(a.value)

    # ---- probes (per-function parity tests) ----
    def intersect(self, rays: np.ndarray) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros(len(rays), abi.HIT_DTYPE)
        check(load().mcrt_probe_intersect(self._h, abi.fptr(rays), len(rays), out.ctypes.data))
        return out

    def trace(self, config: Config, rays: np.ndarray, depth: int = 0) -> np.ndarray:
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 6)
        out = np.zeros((len(rays), 4), np.float32)
        c = config.to_c()
        check(load().mcrt_probe_trace(self._h, C.byref(c), abi.fptr(rays), len(rays), depth, abi.fptr(out)))
        return out


def unpack_rows_device(config: Config, first: int, step: int, packed_ptr: int, frame_ptr: int, stream: int = 0) -> None:
    c = config.to_c()
    check(load().mcrt_unpack_rows_device(C.byref(c), first, step, C.c_void_p(packed_ptr), C.c_void_p(frame_ptr), C.c_void_p(stream)))


def assemble_frame_device(config: Config, world: int, gathered_ptr: int, rank_stride_pixels: int, frame_ptr: int,
                          stream: int = 0) -> None:
    """All ranks' packed rows (one contiguous [world, packed_rows, W, 4] buffer) → the frame, one launch."""
    c = config.to_c()
    check(load().mcrt_assemble_frame_device(C.byref(c), world, C.c_void_p(gathered_ptr), rank_stride_pixels,
                                            C.c_void_p(frame_ptr), C.c_void_p(stream)))


def quantize_rgba8_device(rgba_ptr: int, out_ptr: int, n_pixels: int, stream: int = 0) -> None:
    check(load().mcrt_quantize_rgba8_device(C.c_void_p(rgba_ptr), C.c_void_p(out_ptr), n_pixels, C.c_void_p(stream)))


def probe_mt_uniform(seeds: Sequence[int], n_draws: int, device: int = 0) -> np.ndarray:
    s = np.asarray(seeds, np.uint32)
    out = np.zeros((len(s), n_draws), np.float32)
    check(load().mcrt_probe_mt_uniform(device, s.ctypes.data_as(C.POINTER(C.c_uint32)), len(s), n_draws, abi.fptr(out)))
    return out


def probe_detmath(op: int, x: np.ndarray, y: Optional[np.ndarray] = None, device: int = 0) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    yy = np.ascontiguousarray(y, np.float32) if y is not None else None
    out = np.zeros_like(x)
    check(load().mcrt_probe_detmath(device, op, abi.fptr(x), abi.fptr(yy) if yy is not None else None, x.size, abi.fptr(out)))
    return out


def probe_detmath_range(op: int, lo_bits: int, hi_bits: int, y0: float = 0.0, device: int = 0) -> int:
    bad = C.c_uint64()
    check(load().mcrt_probe_detmath_range(device, op, lo_bits, hi_bits, C.c_float(y0), C.byref(bad)))
    return int(bad.value)


def probe_div_const(d_first: int, d_count: int, two_corrections: bool = False, device: int = 0, mode: Optional[int] = None):
    """(mismatches, a failing divisor or 0): rt::div_frame against the general division, exhaustively (mcrt.h).
    ``mode``: 0 the adopted form, 1 with a second correction, 2 the uncorrected product (the probe's own check: must
    mismatch), 3 rt::sqrt_pos against sqrtf, 4 the device's 1.0f / d against the host's reciprocal the kernels are given."""
    bad, which = C.c_uint64(), C.c_uint32()
    m = (1 if two_corrections else 0) if mode is None else int(mode)
    check(load().mcrt_probe_div_const(device, d_first, d_count, m, C.byref(bad), C.byref(which)))
    return int(bad.value), int(which.value)


def flatten(scene) -> bytes:
    d = _as_desc(scene)
    lib = load()
    n = lib.mcrt_scene_flatten(d.ptr, None, 0)
    if n == 0:
        raise McrtError(abi.MCRT_ERR_INVALID, lib.mcrt_last_error().decode("utf-8", "replace"))
    buf = C.create_string_buffer(n)
    lib.mcrt_scene_flatten(d.ptr, buf, n)
    return buf.raw
