"""Per-rank render throughput of a 1/N shard of the 1080p frame vs frames in flight (predicts N-GPU scaling)."""
import os, sys, time
if len(sys.argv) > 3 and sys.argv[3] != "default":
    os.environ["GPU_MAX_HW_QUEUES"] = sys.argv[3]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes
N = int(sys.argv[1]); F = int(sys.argv[2])
cfg = M.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
sd = scenes.skin_scene("S64", 0)
sc = [M.DeviceScene(sd) for _ in range(F)]
for s in sc: s.set_lanes(1)
st = [torch.cuda.Stream() for _ in range(F)]
rows = sc[0].owned_pixel_rows(cfg, 0, N)
out = [torch.empty((rows, 1920, 4), dtype=torch.float32, device="cuda") for _ in range(F)]
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for k in range(n):
        i = k % F
        sc[i].render_device(cfg, out[i].data_ptr(), 0, N, abi.LAYOUT_PACKED if N > 1 else abi.LAYOUT_FRAME, st[i].cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
run(30)
us = run(600)
print(f"shard 1/{N} F={F} hwq={os.environ.get('GPU_MAX_HW_QUEUES','default')}: {us:.1f} us per shard-frame -> {N}-GPU ceiling {1920*1080/us:.0f} Mpix/s")
