"""ctypes mirror of include/mcrt.h (struct layouts + helpers to build scene descriptions).

Pure declarations: this module loads no library.  The product loader is ``_lib.py``; the test-only
oracle / compiled-reference loaders live under ``tests/``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)

MCRT_OK = 0
MCRT_ERR_INVALID = 1
MCRT_ERR_NO_DEVICE = 2
MCRT_ERR_HIP = 3
MCRT_ERR_NOMEM = 4
LAYOUT_FRAME = 0
LAYOUT_PACKED = 1


class McrtConfig(C.Structure):
    """RayTracer::Config — /root/reference/src/raytracer/raytracer.h:10-38."""

    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("max_bounces", C.c_int32),
        ("samples_per_pixel", C.c_int32),
        ("tile_size", C.c_int32),
        ("thread_count", C.c_int32),
        ("soft_shadows", C.c_int32),
        ("shadow_samples", C.c_int32),
        ("ao_enabled", C.c_int32),
        ("ao_samples", C.c_int32),
        ("ao_radius", C.c_float),
        ("ao_intensity", C.c_float),
        ("dof_enabled", C.c_int32),
        ("aperture", C.c_float),
        ("focus_distance", C.c_float),
        ("gradient_bg", C.c_int32),
        ("gradient_scale", C.c_float),
        ("bg_center", C.c_float * 4),
        ("bg_edge", C.c_float * 4),
    ]


class McrtTexture(C.Structure):
    _fields_ = [
        ("width", C.c_int32),
        ("height", C.c_int32),
        ("n_pixels", C.c_int64),
        ("rgba", c_float_p),
    ]


class McrtMesh(C.Structure):
    _fields_ = [
        ("n_triangles", C.c_int32),
        ("tri_vertices", c_float_p),
        ("tri_texture", c_int32_p),
        ("n_local_triangles", C.c_int32),
        ("local_tri_vertices", c_float_p),
        ("is_outer_layer", C.c_int32),
        ("has_rotation", C.c_int32),
        ("pivot", C.c_float * 3),
        ("rot_x", C.c_float),
        ("rot_z", C.c_float),
    ]


class McrtSceneDesc(C.Structure):
    _fields_ = [
        ("n_meshes", C.c_int32),
        ("meshes", C.POINTER(McrtMesh)),
        ("n_textures", C.c_int32),
        ("textures", C.POINTER(McrtTexture)),
        ("light_position", C.c_float * 3),
        ("light_color", C.c_float * 4),
        ("light_intensity", C.c_float),
        ("light_radius", C.c_float),
        ("camera_position", C.c_float * 3),
        ("camera_target", C.c_float * 3),
        ("camera_up", C.c_float * 3),
        ("camera_fov", C.c_float),
        ("background_color", C.c_float * 4),
    ]


class McrtTile(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class McrtHit(C.Structure):
    _fields_ = [
        ("hit", C.c_int32),
        ("t", C.c_float),
        ("point", C.c_float * 3),
        ("normal", C.c_float * 3),
        ("texture_color", C.c_float * 4),
        ("is_outer_layer", C.c_int32),
    ]


class McrtTimings(C.Structure):
    _fields_ = [
        ("flatten_ms", C.c_float),
        ("h2d_ms", C.c_float),
        ("kernel_ms", C.c_float),
        ("d2h_ms", C.c_float),
        ("total_ms", C.c_float),
    ]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_int, C.c_int, C.c_void_p)

HIT_DTYPE = np.dtype(
    [
        ("hit", np.int32),
        ("t", np.float32),
        ("point", np.float32, 3),
        ("normal", np.float32, 3),
        ("texture_color", np.float32, 4),
        ("is_outer_layer", np.int32),
    ]
)
assert HIT_DTYPE.itemsize == C.sizeof(McrtHit)


# ------------------------------------------------------------------------------------------------
# Python-side value types mirroring the reference's data model (Scene/Mesh/Light/Camera/Config)
# ------------------------------------------------------------------------------------------------
@dataclass
class Config:
    """RayTracer::Config with the reference's in-class defaults (raytracer.h:10-38)."""

    width: int = 256
    height: int = 256
    maxBounces: int = 3
    samplesPerPixel: int = 1
    tileSize: int = 32
    threadCount: int = 0
    softShadows: bool = True
    shadowSamples: int = 8
    aoEnabled: bool = False
    aoSamples: int = 8
    aoRadius: float = 3.0
    aoIntensity: float = 0.5
    dofEnabled: bool = False
    aperture: float = 0.5
    focusDistance: float = 0.0
    gradientBg: bool = True
    gradientScale: float = 1.0
    bgCenter: Sequence[float] = (0.91, 0.89, 0.86, 1.0)
    bgEdge: Sequence[float] = (0.56, 0.63, 0.71, 1.0)

    def to_c(self) -> McrtConfig:
        c = McrtConfig()
        c.width, c.height = int(self.width), int(self.height)
        c.max_bounces = int(self.maxBounces)
        c.samples_per_pixel = int(self.samplesPerPixel)
        c.tile_size = int(self.tileSize)
        c.thread_count = int(self.threadCount)
        c.soft_shadows = int(bool(self.softShadows))
        c.shadow_samples = int(self.shadowSamples)
        c.ao_enabled = int(bool(self.aoEnabled))
        c.ao_samples = int(self.aoSamples)
        c.ao_radius = float(self.aoRadius)
        c.ao_intensity = float(self.aoIntensity)
        c.dof_enabled = int(bool(self.dofEnabled))
        c.aperture = float(self.aperture)
        c.focus_distance = float(self.focusDistance)
        c.gradient_bg = int(bool(self.gradientBg))
        c.gradient_scale = float(self.gradientScale)
        for i in range(4):
            c.bg_center[i] = float(self.bgCenter[i])
            c.bg_edge[i] = float(self.bgEdge[i])
        return c


@dataclass
class Texture:
    """TextureRegion (skin/texture_region.h:8-27). pixels: (n, 4) float32, row-major."""

    width: int
    height: int
    pixels: np.ndarray

    @staticmethod
    def solid(color: Sequence[float], w: int, h: int) -> "Texture":
        px = np.tile(np.asarray(color, dtype=np.float32).reshape(1, 4), (w * h, 1))
        return Texture(w, h, px)


@dataclass
class Mesh:
    """Mesh (scene/mesh.h:12-27) reduced to what the ray tracer reads."""

    triangles: np.ndarray  # (n, 9) float32  v0 v1 v2
    tri_texture: List[Optional[Texture]]  # per triangle; None = nullptr
    isOuterLayer: bool = False
    hasRotation: bool = False
    pivot: Sequence[float] = (0.0, 0.0, 0.0)
    rotX: float = 0.0
    rotZ: float = 0.0
    localTriangles: Optional[np.ndarray] = None  # (n, 9) float32


@dataclass
class Scene:
    """Scene/Light/Camera (scene/scene.h:10-34)."""

    meshes: List[Mesh] = field(default_factory=list)
    light_position: Sequence[float] = (0.0, 0.0, 0.0)
    light_color: Sequence[float] = (0.0, 0.0, 0.0, 1.0)  # Color() default
    light_intensity: float = 1.0
    light_radius: float = 3.0
    camera_position: Sequence[float] = (0.0, 0.0, 0.0)
    camera_target: Sequence[float] = (0.0, 0.0, 0.0)
    camera_up: Sequence[float] = (0.0, 0.0, 0.0)
    camera_fov: float = 60.0
    backgroundColor: Sequence[float] = (0.0, 0.0, 0.0, 1.0)


class SceneDescHolder:
    """Owns the numpy buffers behind a McrtSceneDesc built from a Python ``Scene``."""

    def __init__(self, scene: Scene):
        self._keep = []
        tex_index = {}
        textures: List[Texture] = []
        meshes_c = (McrtMesh * max(1, len(scene.meshes)))()
        for i, m in enumerate(scene.meshes):
            tri = np.ascontiguousarray(m.triangles, dtype=np.float32).reshape(-1, 9)
            loc = (
                np.ascontiguousarray(m.localTriangles, dtype=np.float32).reshape(-1, 9)
                if m.localTriangles is not None
                else np.zeros((0, 9), np.float32)
            )
            tix = np.full(len(tri), -1, dtype=np.int32)
            for t, tex in enumerate(m.tri_texture):
                if tex is None:
                    continue
                key = id(tex)
                if key not in tex_index:
                    tex_index[key] = len(textures)
                    textures.append(tex)
                tix[t] = tex_index[key]
            self._keep += [tri, loc, tix]
            mc = meshes_c[i]
            mc.n_triangles = len(tri)
            mc.tri_vertices = tri.ctypes.data_as(c_float_p)
            mc.tri_texture = tix.ctypes.data_as(c_int32_p)
            mc.n_local_triangles = len(loc)
            mc.local_tri_vertices = loc.ctypes.data_as(c_float_p)
            mc.is_outer_layer = int(bool(m.isOuterLayer))
            mc.has_rotation = int(bool(m.hasRotation))
            for k in range(3):
                mc.pivot[k] = float(m.pivot[k])
            mc.rot_x = float(m.rotX)
            mc.rot_z = float(m.rotZ)
        tex_c = (McrtTexture * max(1, len(textures)))()
        for i, t in enumerate(textures):
            px = np.ascontiguousarray(t.pixels, dtype=np.float32).reshape(-1, 4)
            self._keep.append(px)
            tex_c[i].width = int(t.width)
            tex_c[i].height = int(t.height)
            tex_c[i].n_pixels = len(px)
            tex_c[i].rgba = px.ctypes.data_as(c_float_p)
        d = McrtSceneDesc()
        d.n_meshes = len(scene.meshes)
        d.meshes = meshes_c
        d.n_textures = len(textures)
        d.textures = tex_c
        for k in range(3):
            d.light_position[k] = float(scene.light_position[k])
            d.camera_position[k] = float(scene.camera_position[k])
            d.camera_target[k] = float(scene.camera_target[k])
            d.camera_up[k] = float(scene.camera_up[k])
        for k in range(4):
            d.light_color[k] = float(scene.light_color[k])
            d.background_color[k] = float(scene.backgroundColor[k])
        d.light_intensity = float(scene.light_intensity)
        d.light_radius = float(scene.light_radius)
        d.camera_fov = float(scene.camera_fov)
        self._keep += [meshes_c, tex_c]
        self.desc = d

    @property
    def ptr(self):
        return C.byref(self.desc)


def desc_to_numpy(desc: McrtSceneDesc) -> dict:
    """Deep-copies a scene description into plain numpy/python data (for comparisons/fixtures)."""
    out = {"meshes": [], "textures": []}
    for i in range(desc.n_textures):
        t = desc.textures[i]
        n = int(t.n_pixels)
        px = np.ctypeslib.as_array(t.rgba, shape=(n * 4,)).copy().reshape(n, 4) if n > 0 else np.zeros((0, 4), np.float32)
        out["textures"].append({"width": int(t.width), "height": int(t.height), "pixels": px})
    for i in range(desc.n_meshes):
        m = desc.meshes[i]
        nt, nl = int(m.n_triangles), int(m.n_local_triangles)
        tri = np.ctypeslib.as_array(m.tri_vertices, shape=(nt * 9,)).copy().reshape(nt, 9) if nt else np.zeros((0, 9), np.float32)
        tix = np.ctypeslib.as_array(m.tri_texture, shape=(nt,)).copy() if nt else np.zeros((0,), np.int32)
        loc = np.ctypeslib.as_array(m.local_tri_vertices, shape=(nl * 9,)).copy().reshape(nl, 9) if nl else np.zeros((0, 9), np.float32)
        out["meshes"].append(
            {
                "triangles": tri,
                "tri_texture": tix,
                "localTriangles": loc,
                "isOuterLayer": int(m.is_outer_layer),
                "hasRotation": int(m.has_rotation),
                "pivot": np.array(list(m.pivot), np.float32),
                "rotX": np.float32(m.rot_x),
                "rotZ": np.float32(m.rot_z),
            }
        )
    for k in ("light_position", "light_color", "camera_position", "camera_target", "camera_up", "background_color"):
        out[k] = np.array(list(getattr(desc, k)), np.float32)
    for k in ("light_intensity", "light_radius", "camera_fov"):
        out[k] = np.float32(getattr(desc, k))
    return out


def scene_from_numpy(d: dict) -> Scene:
    """Inverse of desc_to_numpy → Python ``Scene`` (textures shared by index)."""
    texs = [Texture(int(t["width"]), int(t["height"]), np.asarray(t["pixels"], np.float32)) for t in d["textures"]]
    meshes = []
    for m in d["meshes"]:
        meshes.append(
            Mesh(
                triangles=np.asarray(m["triangles"], np.float32),
                tri_texture=[texs[i] if i >= 0 else None for i in np.asarray(m["tri_texture"]).tolist()],
                isOuterLayer=bool(m["isOuterLayer"]),
                hasRotation=bool(m["hasRotation"]),
                pivot=tuple(float(x) for x in m["pivot"]),
                rotX=float(m["rotX"]),
                rotZ=float(m["rotZ"]),
                localTriangles=np.asarray(m["localTriangles"], np.float32),
            )
        )
    return Scene(
        meshes=meshes,
        light_position=tuple(float(x) for x in d["light_position"]),
        light_color=tuple(float(x) for x in d["light_color"]),
        light_intensity=float(d["light_intensity"]),
        light_radius=float(d["light_radius"]),
        camera_position=tuple(float(x) for x in d["camera_position"]),
        camera_target=tuple(float(x) for x in d["camera_target"]),
        camera_up=tuple(float(x) for x in d["camera_up"]),
        camera_fov=float(d["camera_fov"]),
        backgroundColor=tuple(float(x) for x in d["background_color"]),
    )


def declare_common(lib, prefix: str) -> None:
    """Declares argtypes/restypes of the render/probe entry points that the product, the oracle and
    the compiled reference export under different prefixes."""
    desc_p = C.POINTER(McrtSceneDesc)
    cfg_p = C.POINTER(McrtConfig)

    def opt(name, restype, argtypes):
        fn = getattr(lib, prefix + name, None)
        if fn is not None:
            fn.restype = restype
            fn.argtypes = argtypes
        return fn

    opt("generate_tiles", C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(McrtTile), C.c_int])
    opt("render_tile", C.c_int, [desc_p, cfg_p, C.POINTER(McrtTile), c_float_p])
    opt("intersect", C.c_int, [desc_p, c_float_p, C.c_int, C.c_void_p])
    opt("intersect_mesh", C.c_int, [desc_p, C.c_int, c_float_p, C.c_int, C.c_void_p])
    opt("trace", C.c_int, [desc_p, cfg_p, c_float_p, C.c_int, C.c_int, C.c_int, c_float_p])
    opt("shade", C.c_int, [desc_p, C.POINTER(McrtHit), c_float_p, c_float_p, C.c_float, c_float_p])
    opt("in_shadow", C.c_int, [desc_p, c_float_p, c_float_p, c_float_p])
    opt("soft_shadow", C.c_float, [desc_p, c_float_p, c_float_p, C.c_int, C.c_uint32])
    opt("ao", C.c_float, [desc_p, c_float_p, c_float_p, C.c_int, C.c_float, C.c_uint32])
    opt("background", None, [desc_p, cfg_p, C.c_float, C.c_float, c_float_p])
    opt("camera_ray", None, [desc_p, C.c_float, C.c_float, C.c_float, c_float_p])
    opt("mt_uniform", None, [C.c_uint32, C.c_int, c_float_p])
    opt("mt_uniform_std", None, [C.c_uint32, C.c_int, c_float_p])
    opt("seed_cast", C.c_uint32, [C.c_float])
    opt("quantize", None, [c_float_p, C.POINTER(C.c_uint8), C.c_size_t])
    opt("build_skin_scene", C.c_int, [C.POINTER(C.c_uint8), C.c_int, C.c_int, c_float_p, C.POINTER(desc_p)])
    opt("build_default_scene", C.c_int, [c_float_p, C.POINTER(desc_p)])
    opt("builtin_pose", C.c_int, [C.c_int, c_float_p])
    opt("scene_desc_free", None, [desc_p])


def fptr(a: np.ndarray):
    return a.ctypes.data_as(c_float_p)
