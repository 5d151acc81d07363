"""Wall time of the drop-in call TileRenderer.render() (host buffers in and out), with its breakdown."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import minecraftskin_raytracer_amd as M
import scenes
sd = scenes.skin_scene("S64", 0)
for (w, h, b, spp) in ((1920, 1080, 4, 4), (3840, 2160, 8, 16)):
    cfg = M.Config(width=w, height=h, maxBounces=b, samplesPerPixel=spp)
    for i in range(4):
        t0 = time.perf_counter()
        img = M.TileRenderer.render(sd, cfg)
        dt = (time.perf_counter() - t0) * 1e3
        print(f"{w}x{h} b{b} spp{spp} call {i}: {dt:.1f} ms  {M.TileRenderer.lastTimings()}")
    t0 = time.perf_counter(); ok = M.render_png(sd, cfg, "/tmp/mcrt_hostpath.png"); print(f"  render_png: {(time.perf_counter()-t0)*1e3:.1f} ms ok={ok} size={os.path.getsize('/tmp/mcrt_hostpath.png')}")
