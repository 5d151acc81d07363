"""Generates tests/golden/*.npz from the COMPILED REFERENCE (oracle/_ref/libmcref.so, built by
oracle/Makefile from /root/reference where it lies).  Run in the dev container only:

    python tools/make_golden.py

Fixtures are data only: inputs (skin kind / pose / config / rays) and the reference's outputs.
The reference's own tests hold no golden images or numeric pixel tables, so these
vectors are what pins the oracle (tests/test_oracle_golden.py) and the HIP path
(tests/test_gpu_golden.py) where /root/reference does not exist.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import minecraftskin_raytracer_amd as M  # noqa: E402
from minecraftskin_raytracer_amd import abi  # noqa: E402
import oraclelib  # noqa: E402
import scenes  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")

RENDER_CASES = [
    # name, skin, pose, config overrides
    ("cfg1_small", "S64", 0, dict(width=64, height=64, maxBounces=1, samplesPerPixel=1)),
    ("hard_shadows", "S64", 0, dict(width=48, height=48, maxBounces=1, samplesPerPixel=1, softShadows=False)),
    ("b4_spp4", "S64", 0, dict(width=64, height=36, maxBounces=4, samplesPerPixel=4)),
    ("b4_spp4_pose6", "S64", 6, dict(width=64, height=36, maxBounces=4, samplesPerPixel=4)),
    ("tile7_pose3", "S64", 3, dict(width=48, height=48, maxBounces=2, samplesPerPixel=2, tileSize=7)),
    ("ao_on", "S64", 0, dict(width=48, height=48, maxBounces=1, samplesPerPixel=1, aoEnabled=True)),
    ("dof_on", "S64", 0, dict(width=48, height=48, maxBounces=1, samplesPerPixel=2, dofEnabled=True)),
    ("flat_bg", "S64", 0, dict(width=48, height=48, maxBounces=2, samplesPerPixel=1, gradientBg=False)),
    ("legacy_pose1", "S32", 1, dict(width=48, height=48, maxBounces=3, samplesPerPixel=2)),
    ("b8_spp16", "S64", 5, dict(width=40, height=40, maxBounces=8, samplesPerPixel=16, tileSize=16)),
    ("default_white", "default", 0, dict(width=48, height=48, maxBounces=3, samplesPerPixel=1)),
]


def scene_for(ref, skin, pose):
    p = ref.builtin_pose(pose)
    d = ref.build_default_scene(p) if skin == "default" else ref.build_skin_scene(M.synthetic_skin(skin), p)
    return abi.SceneDescHolder(abi.scene_from_numpy(d)), d


def main():
    if not oraclelib.Reference.available():
        raise SystemExit("oracle/_ref/libmcref.so is missing: run `make -C oracle` in the dev container")
    ref = oraclelib.Reference()
    orc = oraclelib.Oracle()
    os.makedirs(OUT, exist_ok=True)

    # (i) RNG: libstdc++ mt19937 + uniform_real_distribution<float>, incl. wrapped-negative seeds
    seeds = np.array([0, 1, 2, 5489, 12345, 0xFFFFFFFF, 0x80000000, 0xFFFFF000, 2463534242, 4294901760, 19650218, 31337,
                      1920 * 32 + 64, 7 * 1920 + 1888, 123456789, 987654321], dtype=np.uint32)
    draws = np.stack([orc.mt_uniform(int(s), 128, std=True) for s in seeds])
    casts_in = np.array([0.0, 1.5, -1.5, 4294967296.0, -4294967296.0, 3.7e9, -3.7e9, 1e12, -1e12, 123456.78, -0.99, 2147483648.0], np.float32)
    casts = np.array([ref.seed_cast(float(f)) for f in casts_in], np.uint32)
    np.savez_compressed(os.path.join(OUT, "rng.npz"), seeds=seeds, draws=draws, cast_in=casts_in, cast_out=casts)

    # (ii)-(iv) per-function vectors on three scenes
    for name, skin, pose in (("S64_pose0", "S64", 0), ("S64_pose6", "S64", 6), ("S32_pose1", "S32", 1)):
        H, d = scene_for(ref, skin, pose)
        rays = scenes.random_rays(1500, seed=100 + pose)
        hits = ref.intersect(H.ptr, rays)
        cfg = abi.Config(maxBounces=2)
        tr = ref.trace(H.ptr, cfg, rays[:400], 0, 2)
        tr_null = ref.trace(H.ptr, None, rays[:200], 0, 1)
        hp = hits[hits["hit"] != 0][:64]
        soft = np.array([ref.soft_shadow(H.ptr, h["point"], h["normal"], 8, 1000 + i) for i, h in enumerate(hp)], np.float32)
        ao = np.array([ref.ao(H.ptr, h["point"], h["normal"], 8, 3.0, 77 + i) for i, h in enumerate(hp)], np.float32)
        view = np.array([0.0, 0.2, 1.0], np.float32)
        shaded = np.stack([ref.shade(H.ptr, h, view) for h in hp]) if len(hp) else np.zeros((0, 4), np.float32)
        uv = np.random.default_rng(3).uniform(0, 1, size=(64, 2)).astype(np.float32)
        bg = np.stack([ref.background(H.ptr, abi.Config(), float(u), float(v)) for u, v in uv])
        cam = np.stack([ref.camera_ray(H.ptr, float(u), float(v), 16.0 / 9.0) for u, v in uv])
        np.savez_compressed(os.path.join(OUT, f"vectors_{name}.npz"), skin=skin, pose=pose, rays=rays, hits=hits, trace=tr,
                            trace_null=tr_null, hit_points=hp, soft=soft, ao=ao, shaded=shaded, uv=uv, background=bg, camera=cam)

    # (v) full renders
    index = []
    for name, skin, pose, kw in RENDER_CASES:
        H, d = scene_for(ref, skin, pose)
        cfg = abi.Config(**kw)
        img = ref.render(H.ptr, cfg)
        np.savez_compressed(os.path.join(OUT, f"render_{name}.npz"), image=img, rgba8=ref.quantize(img).reshape(img.shape))
        index.append({"name": name, "skin": skin, "pose": pose, "config": kw})
    json.dump(index, open(os.path.join(OUT, "renders.json"), "w"), indent=1)

    # scene builder: full description for two scenes (posed, legacy) — pins MeshBuilder/SkinParser
    for name, skin, pose in (("S64_pose6", "S64", 6), ("S32_pose1", "S32", 1)):
        _, d = scene_for(ref, skin, pose)
        flat = {}
        for i, m in enumerate(d["meshes"]):
            for k, v in m.items():
                flat[f"mesh{i}_{k}"] = np.asarray(v)
        for i, t in enumerate(d["textures"]):
            flat[f"tex{i}_wh"] = np.array([t["width"], t["height"]], np.int32)
            flat[f"tex{i}_px"] = t["pixels"]
        for k in ("light_position", "light_color", "camera_position", "camera_target", "camera_up", "background_color",
                  "light_intensity", "light_radius", "camera_fov"):
            flat[k] = np.asarray(d[k])
        flat["n_meshes"] = np.int32(len(d["meshes"]))
        flat["n_textures"] = np.int32(len(d["textures"]))
        np.savez_compressed(os.path.join(OUT, f"scene_{name}.npz"), **flat)
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"wrote {len(os.listdir(OUT))} files, {total / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
