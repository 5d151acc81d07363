"""Ablation timings of the trace kernel on one GPU (development aid, not part of the bench)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes

def t(scene_desc, iters=5, **kw):
    cfg = M.Config(**kw)
    ds = M.DeviceScene(scene_desc)
    frame = torch.empty((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    ds.time_render_device(cfg, frame.data_ptr(), 2)
    r = k = ds.time_render_device(cfg, frame.data_ptr(), iters)
    ds.close()
    return round(r, 4), round(k, 4)

base = dict(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
s64 = scenes.skin_scene("S64", 0)
empty = M.SceneDesc(scenes.simple_scene(cam_pos=(0, 18, 50)))
default = M.MeshBuilder.buildDefaultScene()
rows = []
def run(name, sd, **kw):
    cfg = dict(base); cfg.update(kw)
    rows.append((name, t(sd, **cfg)))
    print(name, rows[-1][1], flush=True)
if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "scale":
        for b in (0, 1, 2, 4):
            run(f"bounces {b}", s64, maxBounces=b)
        for sh in (2, 4, 8, 16):
            run(f"b0 shadowSamples {sh}", s64, maxBounces=0, shadowSamples=sh)
        run("b0 hard", s64, maxBounces=0, softShadows=False)
        run("default white b0 hard", default, maxBounces=0, softShadows=False)
        run("default white b4", default)
        sys.exit(0)
    if len(sys.argv) > 1 and sys.argv[1] == "short":
        run("base S64", s64)
        run("hard shadows", s64, softShadows=False)
        run("bounces 0", s64, maxBounces=0)
        run("empty scene spp4", empty)
        run("4k b8 spp16", s64, width=3840, height=2160, maxBounces=8, samplesPerPixel=16)
        sys.exit(0)
    run("base S64", s64)
    run("S64 pose6", scenes.skin_scene("S64", 6))
    run("hard shadows", s64, softShadows=False)
    run("bounces 0", s64, maxBounces=0)
    run("bounces 0 hard", s64, maxBounces=0, softShadows=False)
    run("spp 1", s64, samplesPerPixel=1)
    run("spp 1 b0 hard", s64, samplesPerPixel=1, maxBounces=0, softShadows=False)
    run("empty scene spp4", empty)
    run("empty scene spp1", empty, samplesPerPixel=1)
    run("default white scene", default)
    run("shadowSamples 2", s64, shadowSamples=2)
    run("tile 16", s64, tileSize=16)
    run("tile 8", s64, tileSize=8)
    run("tile 64", s64, tileSize=64)
    run("4k b8 spp16", s64, width=3840, height=2160, maxBounces=8, samplesPerPixel=16)
