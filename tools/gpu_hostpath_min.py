"""Minimal one-shot render loop (profiling target)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import minecraftskin_raytracer_amd as M, scenes
sd = scenes.skin_scene("S64", 0)
cfg = M.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
host = np.ones((1080, 1920, 4), np.float32)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    t0 = time.perf_counter(); M.TileRenderer.render(sd, cfg, out=host); print(i, round((time.perf_counter() - t0) * 1e3, 3), flush=True)
