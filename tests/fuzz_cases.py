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
    return M.SceneDesc(sc), abi.Config(**kw), f"seed {seed}: {kind} pose {pose} {kw}"
