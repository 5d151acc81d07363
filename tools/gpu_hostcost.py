"""Host cost of enqueueing one render (small frames: the GPU is never the bottleneck of the loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes
ds = M.DeviceScene(scenes.skin_scene("S64", 0)); ds.set_lanes(1)
st = torch.cuda.Stream()
for (w, h, first, step) in ((64, 64, 0, 1), (1920, 1080, 0, 8), (1920, 1080, 0, 1)):
    cfg = M.Config(width=w, height=h, maxBounces=4, samplesPerPixel=4)
    out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    lay = abi.LAYOUT_FRAME if step == 1 else abi.LAYOUT_PACKED
    for _ in range(20):
        ds.render_device(cfg, out.data_ptr(), first, step, lay, st.cuda_stream)
    torch.cuda.synchronize()
    # short burst on an idle queue: pure host cost of the launches of a render
    torch.cuda.synchronize()
    tb = time.perf_counter()
    for _ in range(8):
        ds.render_device(cfg, out.data_ptr(), first, step, lay, st.cuda_stream)
    burst = (time.perf_counter() - tb) / 8 * 1e6
    torch.cuda.synchronize()
    print(f"{w}x{h} shard 1/{step}: burst of 8 on an idle queue: {burst:.1f} us per render (host only)")
    n = 500
    t0 = time.perf_counter()
    for _ in range(n):
        ds.render_device(cfg, out.data_ptr(), first, step, lay, st.cuda_stream)
    t_enq = (time.perf_counter() - t0) / n * 1e6
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n * 1e6
    print(f"{w}x{h} shard 1/{step}: enqueue {t_enq:.1f} us per render, enqueue+drain {t_all:.1f} us per render")
