"""The one-shot host path at 8K, call by call: fresh Image per call (dropped / kept) against one reused buffer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MCRT_WORKSPACE_MB", os.environ.get("WS_MB", "32768"))
import numpy as np
import minecraftskin_raytracer_amd as M
import scenes
sd = scenes.skin_scene("S32", 0)
cfg = M.Config(width=7680, height=4320, maxBounces=8, samplesPerPixel=int(os.environ.get("SPP", "8")))
def call(label, **kw):
    t0 = time.perf_counter()
    img = M.TileRenderer.render(sd, cfg, **kw)
    dt = (time.perf_counter() - t0) * 1e3
    t = M.TileRenderer.lastTimings()
    print(f"{label:28s} wall {dt:8.1f} ms  split " + " ".join(f"{k}={v:.1f}" for k, v in t.items()), flush=True)
    return img
host = np.zeros((cfg.height, cfg.width, 4), np.float32)
for i in range(3): call("reused", out=host)
for i in range(5): call("fresh, dropped"); 
kept = []
for i in range(3): kept.append(call("fresh, kept"))
del kept
for i in range(3): call("reused again", out=host)
for i in range(3): call("fresh, dropped again")
