"""Do two renders on two streams overlap?  (hypothesis test for multi-stream batches)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes

cfg = M.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
sd = scenes.skin_scene("S64", 0)
sc = [M.DeviceScene(sd), M.DeviceScene(sd)]
frames = [torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]

def run(n, two, shard):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        if two:
            for k in range(2):
                if shard:   # the two halves of ONE frame (cyclic tile rows), packed
                    sc[k].render_device(cfg, frames[k].data_ptr(), k, 2, abi.LAYOUT_PACKED, streams[k].cuda_stream)
                else:       # two whole frames
                    sc[k].render_device(cfg, frames[k].data_ptr(), 0, 1, abi.LAYOUT_FRAME, streams[k].cuda_stream)
        else:
            sc[0].render_device(cfg, frames[0].data_ptr(), 0, 1, abi.LAYOUT_FRAME, streams[0].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

for _ in range(2):
    run(10, False, False); run(10, True, False); run(10, True, True)
print("one frame, one stream        ms/iter", round(run(200, False, False), 4))
print("two frames, two streams      ms/iter", round(run(200, True, False), 4), "(per frame: half)")
print("one frame as 2 shards/streams ms/iter", round(run(200, True, True), 4))
