"""Scene fixtures shared by the tests: hand-built boxes in the style of the reference's unit tests
(/root/reference/tests/test_intersection.cpp:6-18, test_shading.cpp, test_raytracer_props.cpp) and
the synthetic skin scenes of SURVEY.md §8(d)."""
from __future__ import annotations

import numpy as np

import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi

f32 = np.float32

_QUADS = [(2, 3, 1, 0), (7, 6, 4, 5), (3, 7, 5, 1), (6, 2, 0, 4), (6, 7, 3, 2), (0, 1, 5, 4)]


def build_box(face_textures, position, size, offset=0.0) -> abi.Mesh:
    """MeshBuilder::buildBox (mesh_builder.cpp:66-123) in float32; face_textures = dict or a single
    Texture for all six faces; order of triangles: back, front, left, right, top, bottom."""
    if isinstance(face_textures, abi.Texture) or face_textures is None:
        face_textures = {k: face_textures for k in ("back", "front", "left", "right", "top", "bottom")}
    px, py, pz = (f32(v) for v in position)
    hw = f32(size[0]) / f32(2.0) + f32(offset)
    hh = f32(size[1]) / f32(2.0) + f32(offset)
    hd = f32(size[2]) / f32(2.0) + f32(offset)
    xs, ys, zs = (px - hw, px + hw), (py - hh, py + hh), (pz - hd, pz + hd)
    corner = [(xs[i & 1], ys[(i >> 1) & 1], zs[(i >> 2) & 1]) for i in range(8)]
    tris, tex = [], []
    for name, q in zip(("back", "front", "left", "right", "top", "bottom"), _QUADS):
        for t in ((q[0], q[1], q[2]), (q[0], q[2], q[3])):
            tris.append([c for k in t for c in corner[k]])
            tex.append(face_textures[name])
    return abi.Mesh(triangles=np.asarray(tris, f32), tri_texture=tex, isOuterLayer=offset > 0.0)


def solid(color, w=4, h=4) -> abi.Texture:
    return abi.Texture.solid(color, w, h)


def simple_scene(meshes=(), light=(0, 50, 50), cam_pos=(0, 18, 40), cam_target=(0, 18, 0), bg=(0.1, 0.1, 0.1, 1.0),
                 light_color=(1, 1, 1, 1), fov=60.0, radius=3.0) -> abi.Scene:
    return abi.Scene(meshes=list(meshes), light_position=light, light_color=light_color, light_intensity=1.0,
                     light_radius=radius, camera_position=cam_pos, camera_target=cam_target, camera_up=(0, 1, 0),
                     camera_fov=fov, backgroundColor=bg)


def skin_scene(kind="S64", pose_index=0) -> "M.SceneDesc":
    return M.MeshBuilder.buildScene(M.synthetic_skin(kind), M.getBuiltinPoses()[pose_index])


def random_rays(n, seed=0, target=(0.0, 18.0, 0.0), spread=12.0, dist=(20.0, 60.0)) -> np.ndarray:
    """Rays aimed at the character from random directions (plus some degenerate ones)."""
    g = np.random.default_rng(seed)
    d = g.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    r = g.uniform(dist[0], dist[1], size=(n, 1))
    o = np.asarray(target) + d * r
    aim = np.asarray(target) + g.uniform(-spread, spread, size=(n, 3)) * np.array([0.7, 1.4, 0.4])
    dd = aim - o
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    rays = np.concatenate([o, dd], axis=1).astype(f32)
    # axis-parallel rays (exercise the |d| < 1e-8 slab branch) and rays starting inside boxes
    k = max(1, n // 16)
    rays[:k, 3:] = np.eye(3, dtype=f32)[g.integers(0, 3, k)] * g.choice([-1.0, 1.0], size=(k, 1)).astype(f32)
    rays[k:2 * k, :3] = (np.asarray(target) + g.uniform(-3, 3, size=(k, 3))).astype(f32)
    return rays


def bits(a) -> np.ndarray:
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def assert_bit_equal(a, b, what=""):
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    neq = bits(a) != bits(b)
    # NaN payloads may differ; treat NaN == NaN
    neq &= ~(np.isnan(a) & np.isnan(b))
    if neq.any():
        idx = np.argwhere(neq)
        first = tuple(idx[0])
        raise AssertionError(f"{what}: {neq.sum()} of {neq.size} floats differ; first at {first}: {a[first]!r} vs {b[first]!r}; max abs diff {np.nanmax(np.abs(a - b))}")


def assert_hits_equal(a, b, what=""):
    assert a.dtype == b.dtype and a.shape == b.shape
    assert np.array_equal(a["hit"], b["hit"]), what + " hit flags"
    m = a["hit"] != 0
    for f in ("t", "point", "normal", "texture_color"):
        assert_bit_equal(a[f][m], b[f][m], f"{what} {f}")
    assert np.array_equal(a["is_outer_layer"][m], b["is_outer_layer"][m]), what + " outer flags"
