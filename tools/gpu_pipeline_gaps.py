"""Where does the time of F frames in flight go?  Host enqueue cost against device time, F handles round-robin."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes
F = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 400
sd = scenes.skin_scene("S64", 0)
cfg = M.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
hs = [M.DeviceScene(sd) for _ in range(F)]
for h in hs: h.set_lanes(1)
sts = [torch.cuda.Stream() for _ in range(F)]
outs = [torch.empty((1080, 1920, 4), dtype=torch.float32, device="cuda") for _ in range(F)]
def step(k):
    s = k % F
    hs[s].render_device(cfg, outs[s].data_ptr(), 0, 1, abi.LAYOUT_FRAME, sts[s].cuda_stream)
for k in range(6 * F): step(k)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(N): step(k)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"F={F}: enqueue {1e6*(t1-t0)/N:.1f} us per frame (host), all done {1e6*(t2-t0)/N:.1f} us per frame; drain after the last enqueue {1e6*(t2-t1):.0f} us")
# burst: host cost on idle queues
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(F): step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"burst of {F} on idle queues: {1e6*(t1-t0)/F:.1f} us per enqueue")
