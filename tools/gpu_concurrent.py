"""One frame rendered as S interleaved tile-row shards on S streams (hypothesis test for lanes)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import minecraftskin_raytracer_amd as M
from minecraftskin_raytracer_amd import abi
import scenes

CASES = {"base": dict(width=1920, height=1080, maxBounces=4, samplesPerPixel=4),
         "4k": dict(width=3840, height=2160, maxBounces=8, samplesPerPixel=16)}
case = sys.argv[1] if len(sys.argv) > 1 else "base"
cfg = M.Config(**CASES[case])
sd = scenes.skin_scene("S64", 0)
NMAX = 8
sc = [M.DeviceScene(sd) for _ in range(NMAX)]
frame = torch.empty((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
ref = torch.empty_like(frame)
streams = [torch.cuda.Stream() for _ in range(NMAX)]
sc[0].render_device(cfg, ref.data_ptr(), 0, 1, abi.LAYOUT_FRAME, streams[0].cuda_stream)
torch.cuda.synchronize()

def run(n, S):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for k in range(S):
            sc[k].render_device(cfg, frame.data_ptr(), k, S, abi.LAYOUT_FRAME, streams[k].cuda_stream)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

iters = 200 if case == "base" else 20
for S in (1, 2, 3, 4, 6, 8):
    run(5, S)
    frame.zero_()
    ms = run(iters, S)
    print(f"{case}: {S} shard(s)/stream(s): {ms:.4f} ms/frame  equal={bool(torch.equal(frame, ref))}")
