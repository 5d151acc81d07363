"""Seeded random (scene, config) cases for the HIP-vs-oracle parity sweep (tools/gpu_fuzz.py, and a
short fixed-seed run in tests/test_gpu_fuzz.py).  Cameras are placed all around the figure (inside
the bounding box too), lights move, the frame/tile geometry is ragged on purpose."""
from __future__ import annotations

import numpy as np

import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi


def make_case(seed: int):
    g = np.random.default_rng(seed)
    kind = ["S64", "S64", "S32"][g.integers(0, 3)]
    pose = int(g.integers(0, len(M.getBuiltinPoses())))
    base = M.MeshBuilder.buildScene(M.synthetic_skin(kind), M.getBuiltinPoses()[pose])
    sc = abi.scene_from_numpy(base.to_numpy())
    r = g.random()
    if r < 0.55:  # orbit at a distance
        ang, elev, dist = g.uniform(0, 2 * np.pi), g.uniform(-0.6, 1.2), g.uniform(18, 90)
        sc.camera_position = (float(dist * np.cos(elev) * np.sin(ang)), float(16 + dist * np.sin(elev)), float(dist * np.cos(elev) * np.cos(ang)))
        sc.camera_target = (float(g.uniform(-3, 3)), float(g.uniform(8, 28)), float(g.uniform(-3, 3)))
    elif r < 0.7:  # close / inside the figure's bounding box: no screen bounds, origins inside boxes
        sc.camera_position = (float(g.uniform(-6, 6)), float(g.uniform(2, 30)), float(g.uniform(-6, 6)))
        sc.camera_target = (float(g.uniform(-10, 10)), float(g.uniform(0, 32)), float(g.uniform(-10, 10)))
    # else: the builder's camera
    if g.random() < 0.5:
        sc.light_position = (float(g.uniform(-40, 40)), float(g.uniform(-10, 60)), float(g.uniform(-40, 40)))
    if g.random() < 0.3:
        sc.light_radius = float([0.0, 0.5, 3.0, 9.0][g.integers(0, 4)])
    if g.random() < 0.3:
        sc.camera_fov = float(g.uniform(20, 110))
    w, h = int(g.integers(9, 150)), int(g.integers(9, 110))
    # 13 spp: more than 24 draws per pixel (the tiles' streams go through HBM); 9 bounces: the general per-level variants
    kw = dict(width=w, height=h, maxBounces=int([0, 1, 2, 3, 4, 5, 6, 3, 4, 9][g.integers(0, 10)]),
              samplesPerPixel=int([1, 1, 2, 3, 4, 7, 4, 13][g.integers(0, 8)]),
              tileSize=int([1, 3, 8, 16, 32, 32, 50, 200][g.integers(0, 8)]))
    if g.random() < 0.25:
        kw["softShadows"] = False
    if g.random() < 0.3:
        kw["shadowSamples"] = int([1, 2, 3, 4, 6, 16, 32][g.integers(0, 7)])
    if g.random() < 0.35:
        kw.update(aoEnabled=True, aoSamples=int([1, 2, 5, 8, 16, 32][g.integers(0, 6)]), aoRadius=float(g.uniform(0.5, 8)),
                  aoIntensity=float(g.uniform(0, 1)))
    if g.random() < 0.35:
        kw.update(dofEnabled=True, aperture=float([1e-7, 0.05, 0.3, 1.0, 4.0][g.integers(0, 5)]),
                  focusDistance=float([0.0, 0.0, 10.0, 45.0, 300.0][g.integers(0, 5)]))
    if g.random() < 0.2:
        kw["gradientBg"] = False
    # (drawn last: the cases before these lines keep their seeds' meaning)  gradient radii from "the whole frame is edge
    # colour" to "no pixel reaches it" — tiles of one colour are filled without samples — and unusual colours
    if g.random() < 0.35:
        kw["gradientScale"] = float([0.1, 0.5, 0.7071, 1.0, 1.4142, 2.0, 5.0, 0.0][g.integers(0, 8)] * g.uniform(0.98, 1.02))
    if g.random() < 0.15:
        kw["bgCenter"] = tuple(float(x) for x in g.uniform(-0.5, 1.5, 4))
        kw["bgEdge"] = tuple(float(x) for x in np.where(g.random(4) < 0.3, 0.0, g.uniform(-0.5, 1.5, 4)))
    return M.SceneDesc(sc), abi.Config(**kw), f"seed {seed}: {kind} pose {pose} {kw}"


# ---------------------------------------------------------------------------------------------------------
# Cases aimed at `lit`'s whole-bundle decisions (rt_core.h: bundle_decide): every hit whose S shadow rays are
# declared all-lit or all-shadowed without being traced must agree with the oracle, which traces them all.
# Lights close to / inside / grazing the boxes, texel grids from 1x1 to 64x64 with every alpha density,
# transparent texels on inner layers, null textures, nested and touching boxes, flat boxes, partly posed
# figures, scenes scaled from 1e-3 to 1e4, light radii from 0 to larger than the figure.
# ---------------------------------------------------------------------------------------------------------
def _random_texture(g, outer: bool):
    w, h = int([1, 1, 2, 3, 4, 8, 8, 16, 64][g.integers(0, 9)]), int([1, 2, 3, 4, 8, 8, 16, 64][g.integers(0, 8)])
    px = g.random((w * h, 4)).astype(np.float32)
    density = [0.0, 0.05, 0.4, 0.9, 1.0][g.integers(0, 5)] if outer else [1.0, 1.0, 1.0, 0.9, 0.5][g.integers(0, 5)]
    alpha = (g.random(w * h) < density).astype(np.float32)
    if g.random() < 0.2:  # alphas other than 0 and 1 (only == 0 and > 0 matter; negative ones are neither)
        alpha = alpha * g.choice(np.asarray([1.0, 0.5, 1e-6, -1.0], np.float32), size=w * h)
    px[:, 3] = alpha
    return abi.Texture(w, h, px)


def _box(g, centre, size, offset, outer):
    import scenes

    faces = {}
    shared = _random_texture(g, outer)
    for name in ("back", "front", "left", "right", "top", "bottom"):
        r = g.random()
        faces[name] = None if r < 0.04 else (_random_texture(g, outer) if r < 0.5 else shared)
    return scenes.build_box(faces, centre, size, offset)


def _box_scene(g):
    meshes = []
    n = int(g.integers(1, 6))
    centres = []
    for _ in range(n):
        c = g.uniform(-8, 8, 3)
        size = g.uniform(0.5, 9, 3)
        if g.random() < 0.08:
            size[g.integers(0, 3)] = 0.0  # a flat box
        if centres and g.random() < 0.3:  # touching / overlapping a previous box
            c = centres[g.integers(0, len(centres))] + g.uniform(-1, 1, 3) * size
        centres.append(c)
        meshes.append(_box(g, c, size, 0.0, False))
        if g.random() < 0.7:  # its outer layer, from a hair's breadth to a wide shell
            meshes.append(_box(g, c, size, float([1e-3, 2e-3, 0.02, 0.25, 0.5, 2.0][g.integers(0, 6)]), True))
    if g.random() < 0.3:  # a floor under everything
        meshes.append(_box(g, (0.0, -14.0, 0.0), (60.0, 1.0, 60.0), 0.0, False))
    centre = np.mean(np.asarray(centres), axis=0)
    return meshes, centre


def make_bundle_case(seed: int, wide: bool = False):
    """wide: the scene is also scaled by 1e-5 ... 1e6 or moved 1e3 ... 1e6 away from the origin (coordinates whose ulp is
    as large as 0.06), with lights much smaller than the scene — the margins of the conservative masks and of the
    decisions are multiples of the scene's coordinate magnitude (flat_scene.h: mask_slack), not absolute numbers."""
    import scenes

    g = np.random.default_rng(seed ^ 0x5EED0000)
    if g.random() < 0.5:
        meshes, centre = _box_scene(g)
        sc = scenes.simple_scene(meshes)
        what = f"boxes x{len(meshes)}"
    else:  # the figure, partly posed with free angles, with its own alpha densities
        kind = ["S64", "S64", "S32"][g.integers(0, 3)]
        skin = M.synthetic_skin(kind, seed=int(g.integers(1, 1 << 30)))
        r = g.random()
        if r < 0.3:
            skin[..., 3] = np.where(skin[..., 3] > 0, 255, np.where(g.random(skin.shape[:2]) < g.random(), 255, 0))
        elif r < 0.4:
            skin[..., 3] = np.where(g.random(skin.shape[:2]) < 0.97, skin[..., 3], 0)  # holes in the inner layer too
        pose = np.zeros(12, np.float32)
        for k in range(12):
            if g.random() < 0.25:
                pose[k] = g.uniform(-1.5, 1.5) if g.random() < 0.8 else g.uniform(-0.02, 0.02)  # around the 0.01 gate too
        base = M.MeshBuilder.buildScene(skin, pose)
        sc = abi.scene_from_numpy(base.to_numpy())
        centre = np.asarray([0.0, 16.0, 0.0])
        what = f"{kind} pose {np.round(pose, 3).tolist()}"
    # light: far, near a box, inside the figure's bounds, grazing a face plane
    r = g.random()
    if r < 0.35:
        lp = centre + g.normal(size=3) * g.uniform(15, 80)
    elif r < 0.7:
        lp = centre + g.uniform(-10, 10, 3) * np.asarray([1.0, 1.6, 1.0])
    elif r < 0.85:
        lp = centre + g.uniform(-3, 3, 3)
    else:  # on the plane of some face, far out
        tri = sc.meshes[g.integers(0, len(sc.meshes))].triangles
        lp = np.asarray(tri[g.integers(0, len(tri))][:3], np.float64) + np.asarray([g.uniform(-40, 40), 0.0, g.uniform(-40, 40)])
    sc.light_position = tuple(float(x) for x in lp)
    sc.light_radius = float([0.0, 1e-3, 0.05, 0.5, 3.0, 3.0, 9.0, 25.0][g.integers(0, 8)])
    sc.light_color = (1.0, 1.0, 1.0, 1.0)
    ang, elev, dist = g.uniform(0, 2 * np.pi), g.uniform(-0.8, 1.2), g.uniform(14, 60)
    sc.camera_position = tuple(float(x) for x in centre + dist * np.asarray([np.cos(elev) * np.sin(ang), np.sin(elev), np.cos(elev) * np.cos(ang)]))
    sc.camera_target = tuple(float(x) for x in centre + g.uniform(-2, 2, 3))
    sc.camera_up = (0.0, 1.0, 0.0)
    # the whole scene scaled (margins of the decisions are relative to the scene's size)
    scale = float([1.0, 1.0, 1.0, 1e-3, 0.05, 30.0, 1e4][g.integers(0, 7)])
    shift = np.zeros(3, np.float32)
    if wide:  # (an own generator: the draws above keep their meaning)
        gw = np.random.default_rng(seed ^ 0x77EED000)
        r = gw.random()
        if r < 0.45:
            scale = float([1e-5, 1e-4, 1e5, 1e6, 3e5][gw.integers(0, 5)])
        elif r < 0.9:
            scale = 1.0
            shift = (gw.choice([1e3, 1e4, 1e5, 1e6, 3e6]) * gw.choice([-1.0, 1.0], 3) * (gw.random(3) < 0.6)).astype(np.float32)
        if gw.random() < 0.4:
            sc.light_radius = float([1e-4, 1e-3, 0.02][gw.integers(0, 3)])  # a light much smaller than the scene
    if scale != 1.0 or shift.any():
        s32 = np.float32(scale)
        for m in sc.meshes:
            m.triangles = (np.asarray(m.triangles, np.float32) * s32 + np.tile(shift, 3)).astype(np.float32)
            if m.localTriangles is not None:
                m.localTriangles = (np.asarray(m.localTriangles, np.float32) * s32 + np.tile(shift, 3)).astype(np.float32)
            m.pivot = tuple(float(np.float32(x) * s32 + t) for x, t in zip(m.pivot, shift))
        sc.light_position = tuple(float(np.float32(x) * s32 + t) for x, t in zip(sc.light_position, shift))
        sc.light_radius = float(np.float32(sc.light_radius) * s32)
        sc.camera_position = tuple(float(np.float32(x) * s32 + t) for x, t in zip(sc.camera_position, shift))
        sc.camera_target = tuple(float(np.float32(x) * s32 + t) for x, t in zip(sc.camera_target, shift))
    w, h = int(g.integers(24, 120)), int(g.integers(24, 90))
    kw = dict(width=w, height=h, maxBounces=int([0, 1, 2, 4][g.integers(0, 4)]), samplesPerPixel=int([1, 2, 4][g.integers(0, 3)]),
              tileSize=int([8, 16, 32, 32][g.integers(0, 4)]), shadowSamples=int([2, 3, 4, 8, 8, 8, 16, 32][g.integers(0, 8)]))
    if g.random() < 0.1:
        kw.update(aoEnabled=True, aoSamples=int([2, 8][g.integers(0, 2)]), aoRadius=float(g.uniform(0.5, 4) * scale))
    return M.SceneDesc(sc), abi.Config(**kw), f"bundle seed {seed}: {what} light {np.round(lp, 3).tolist()} r {sc.light_radius} scale {scale} shift {shift.tolist()} {kw}"


def make_wide_case(seed: int):
    return make_bundle_case(seed, wide=True)
