"""Lane count x workspace budget, back-to-back and synced.  MCRT_LANES / MCRT_WORKSPACE_MB are read
once per process, so each configuration runs in a child process.
usage: gpu_lanes.py <case> <lanes,lanes,...> <budget_mb,budget_mb,...>"""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import minecraftskin_raytracer_amd as M
    from minecraftskin_raytracer_amd import abi
    import scenes
    case = sys.argv[2]
    kw = {"base": dict(width=1920, height=1080, maxBounces=4, samplesPerPixel=4),
          "4k": dict(width=3840, height=2160, maxBounces=8, samplesPerPixel=16),
          "4k4": dict(width=3840, height=2160, maxBounces=4, samplesPerPixel=4)}[case]
    cfg = M.Config(**kw)
    ds = M.DeviceScene(scenes.skin_scene("S64", 0))
    frame = torch.empty((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    h = torch.cuda.current_stream().cuda_stream
    n = 200 if case == "base" else 20
    for _ in range(3):
        ds.render_device(cfg, frame.data_ptr(), 0, 1, abi.LAYOUT_FRAME, h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ds.render_device(cfg, frame.data_ptr(), 0, 1, abi.LAYOUT_FRAME, h)
    torch.cuda.synchronize()
    thr = (time.perf_counter() - t0) / n * 1e3
    t0 = time.perf_counter()
    for _ in range(n):
        ds.render_device(cfg, frame.data_ptr(), 0, 1, abi.LAYOUT_FRAME, h)
        torch.cuda.synchronize()
    lat = (time.perf_counter() - t0) / n * 1e3
    print(f"{case} lanes={os.environ.get('MCRT_LANES','auto')} split={os.environ.get('MCRT_LANE_SPLIT','1')} stagger={os.environ.get('MCRT_LANE_STAGGER','0')}: back-to-back {thr:.4f} ms/frame, synced {lat:.4f} ms/frame")
    sys.exit(0)
case = sys.argv[1]
extra = dict(kv.split("=") for kv in sys.argv[4:])
for budget in sys.argv[3].split(","):
    for lanes in sys.argv[2].split(","):
        env = dict(os.environ, MCRT_LANES=lanes, MCRT_WORKSPACE_MB=budget, **extra)
        r = subprocess.run([sys.executable, __file__, "child", case], env=env, capture_output=True, text=True, timeout=300)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
