#!/usr/bin/env python3
"""bench.py — throughput of the tile-render hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--frames-in-flight F]

A *step* is one render of the BASELINE.json north-star frame (configs[1]: 1920x1080, 4 bounces,
4 samples/pixel, one light, synthetic 64x64 skin with inner+outer layer, `RayTracer::Config`
defaults otherwise) with the flattened scene already resident in HBM; the output is the float4
framebuffer in HBM.  Three figures describe it, all in the line:

  value / ms_per_step   DEVICE THROUGHPUT with F frames (default 4) in flight: step k uses scene handle / stream /
                        output buffer k mod F, every frame is rendered completely (`--check` compares each buffer
                        with a lone render).  This is the contract's `value` (inputs resident in HBM).
  latency_ms            one frame at a time on the device: enqueue, wait, repeat (library defaults: the render
                        spreads itself over 2 internal streams); kernel.pipeline_ms is the same frame on ONE
                        stream, measured with hipEvents on that stream.
  render_call           SURVEY.md §8(d)'s metric — THE CONTRACT'S t: wall time of ONE `TileRenderer::render()` call
                        (scene flatten + upload + kernels + download into host memory + progress callbacks) returning a
                        FRESH Image, as the reference's call site gets one (tile_renderer.cpp:141), median of >= 10 calls
                        after 2 warm-ups, with the library's own split; beside it the same call into a reused buffer, the
                        first call of a cold process, and the C++ drop-in binary (tools/micro/dropin_time.cpp).

N > 1 (launched by torch.distributed.run, one process per GPU): the SAME frame is sharded by cyclic tile
rows (rank r renders tile rows r, r+N, ...), each rank renders into a packed buffer and an RCCL gather over
xGMI assembles the frame on rank 0 (strong scaling: the total work is fixed); gathers overlap the renders of
the following steps.

Rank 0 prints ONE JSON line: metric/value per BASELINE.json plus `roofline` — algorithmic bytes (16 B per
output pixel) over the hipEvent-measured duration of one frame's pipeline against the HBM peak, the HBM
traffic and the VALU wave-instructions per frame from the committed PMC passes (profiles/pmc_traffic.json),
and from those the VALU fraction against 78.6 T lane-ops/s: this path is VALU-bound, see DESIGN.md — and, at
N = 1, `cpu_baseline` (the compiled reference, or the oracle port when oracle/_ref is absent, timed on this
box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The frames in flight run on one stream each; the HIP runtime multiplexes streams onto 4 hardware
# queues by default, and 4 streams sharing 4 queues with the gather's stream serialise.  Must be set
# before the runtime initialises (first torch.cuda / HIP call).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec
VALU_PEAK_TLOPS = 78.6  # 157.3 TFLOP/s fp32 vector = 78.6 T non-FMA lane-ops/s

WORKLOADS = {
    # name: (width, height, bounces, spp, skin, pose[, extra Config fields])
    "1080p_b4_spp4_S64": (1920, 1080, 4, 4, "S64", 0),  # BASELINE.json configs[1] — the metric's config
    "4k_b8_spp16_S64": (3840, 2160, 8, 16, "S64", 0),  # configs[2]
    "4k_b4_spp4_S64": (3840, 2160, 4, 4, "S64", 0),  # configs[3]
    "8k_b8_spp64_S32": (7680, 4320, 8, 64, "S32", 0),  # configs[4] (single reference light)
    "256_b1_spp1_S64": (256, 256, 1, 1, "S64", 0),  # configs[0]
    # what the reference GUI renders by default (main_window.cpp:242-348 as the reference states it):
    # 1920x1080, 4 bounces, 64 spp, AO 16 samples, depth of field aperture 0.3, soft shadows 8
    "gui_defaults": (1920, 1080, 4, 64, "S64", 0, dict(aoEnabled=True, aoSamples=16, dofEnabled=True, aperture=0.3)),
}


def workload_config(M, name: str):
    w, h, b, spp, skin, pose, *extra = WORKLOADS[name]
    return M.Config(width=w, height=h, maxBounces=b, samplesPerPixel=spp, **(extra[0] if extra else {})), skin, pose


def kernel_source_hash() -> str:
    """Identifies the kernels a PMC pass was taken on: profiles/pmc_traffic.json entries carry it, and a bench run on
    different sources flags its counters as stale."""
    import hashlib

    h = hashlib.sha1()
    for rel in ("minecraftskin_raytracer_amd/csrc/render_kernels.hip", "minecraftskin_raytracer_amd/csrc/rt_core.h",
                "minecraftskin_raytracer_amd/csrc/kernels.h", "minecraftskin_raytracer_amd/csrc/flat_scene.h", "include/mcrt_detmath.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:12]


def cpu_baseline(workload: str, gpu_frame=None, frames: int = 3, budget_s: float = 30.0) -> dict:
    """Times the reference's own std::thread TileRenderer (oracle/_ref) — or the oracle port — on the host cores of
    this box, same scene/config, threadCount = 0 (all cores), and compares what it rendered with the GPU frame
    (north_star's parity sentence at full size, with this host's libm).  Frames the CPU cannot render whole within
    the budget are timed on a cyclic sample of their tile rows (every k-th row, the reference's renderTile on the
    same thread pool) and scaled by rows / sampled rows — stated in `sample`."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oraclelib
    import minecraftskin_raytracer_amd as M

    cfg, skin, pose = workload_config(M, workload)
    w, h = cfg.width, cfg.height
    kind = "reference" if oraclelib.Reference.available() else "port"
    lib = oraclelib.Reference() if kind == "reference" else oraclelib.Oracle()
    sd = M.MeshBuilder.buildScene(M.synthetic_skin(skin), M.getBuiltinPoses()[pose])
    tiles_y = (h + cfg.tileSize - 1) // cfg.tileSize
    # ~7 M samples/s on this class of host at the metric frame (8.3 M samples in 1.1 s); AO multiplies the work per hit
    samples = w * h * max(cfg.samplesPerPixel, 1) * (3 if cfg.aoEnabled else 1)
    step = max(1, int(round(samples / (7.0e6 * budget_s / 2))))
    step = min(step, max(1, tiles_y // 4))  # at least four sampled rows
    times = []
    t_all = time.perf_counter()
    if step == 1:
        for _ in range(frames):
            t0 = time.perf_counter()
            img = lib.render(sd.ptr, cfg)
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_all > budget_s:
                break
        med = statistics.median(times)
        rows = list(range(tiles_y))
        sample = f"{len(times)} full frame(s) of {workload}, median; threadCount=0 (std::thread pool over all host cores)"
    else:
        first = step // 2
        rows = list(range(first, tiles_y, step))
        img = np.zeros((h, w, 4), np.float32)
        t0 = time.perf_counter()
        lib.render_rows(sd.ptr, cfg, first, step, img)
        part = time.perf_counter() - t0
        times.append(part)
        med = part * tiles_y / len(rows)
        sample = (f"{len(rows)} of {tiles_y} tile rows of {workload} (rows {first}, {first + step}, ...: every {step}th), rendered once by "
                  f"renderTile on the same thread pool in {part:.2f} s and scaled by {tiles_y}/{len(rows)}; threadCount=0")
    out = {
        "value": round(w * h / med / 1e6, 4),
        "unit": "Mpixels/s",
        "ms_per_frame": round(med * 1e3, 2),
        "cores": os.cpu_count(),
        "kind": kind,
        "sample": sample,
    }
    if gpu_frame is not None:
        T = cfg.tileSize
        px = np.concatenate([np.arange(r * T, min(h, (r + 1) * T)) for r in rows])
        a, b = img[px], gpu_frame[px]
        out["frame_equals_gpu"] = bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
        out["rgba8_equals_gpu"] = bool(np.array_equal(M.quantize_rgba8(a), M.quantize_rgba8(b)))
        out["compared_pixels"] = int(a.shape[0] * a.shape[1])
        if not out["frame_equals_gpu"]:
            out["differing_floats"] = int((a.view(np.uint32) != b.view(np.uint32)).sum())
            out["max_abs_diff"] = float(np.nanmax(np.abs(a - b)))
    return out


COLD_CALL = r"""
import json, sys, time
sys.path.insert(0, {root!r})
t0 = time.perf_counter()
import numpy as np
import minecraftskin_raytracer_amd as M
sys.path.insert(0, {root!r})
import bench
cfg, skin, pose = bench.workload_config(M, {workload!r})
sd = M.MeshBuilder.buildScene(M.synthetic_skin(skin), M.getBuiltinPoses()[pose])
t1 = time.perf_counter()
M.TileRenderer.render(sd, cfg)
t2 = time.perf_counter()
err = M.TileRenderer.lastErrors()
M.TileRenderer.render(sd, cfg)
t3 = time.perf_counter()
print(json.dumps({{"first_ms": round((t2 - t1) * 1e3, 2), "second_ms": round((t3 - t2) * 1e3, 3), "import_and_scene_ms": round((t1 - t0) * 1e3, 1), "errors": err}}))
"""


def cold_call(workload: str):
    """The first TileRenderer::render() of a fresh process (HIP start-up, code load, workspace allocation, seed table):
    what one button press of the reference GUI pays once."""
    import subprocess

    try:
        p = subprocess.run([sys.executable, "-c", COLD_CALL.format(root=ROOT, workload=workload)], capture_output=True, text=True, timeout=300)
        return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def cpp_dropin(cfg) -> dict:
    """The C++ drop-in (csrc/host/tile_renderer_hip.cpp behind the reference's TileRenderer interface) timed by its own
    binary: a reference-shaped Scene in, a fresh Image out per call (tools/micro/dropin_time.cpp)."""
    import subprocess
    import tempfile

    pkg = os.path.join(ROOT, "minecraftskin_raytracer_amd")
    host = os.path.join(pkg, "csrc", "host")
    try:
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "dropin_time")
            subprocess.check_call(["g++", "-std=c++17", "-O2", f"-I{ROOT}/include", f"-I{host}", os.path.join(ROOT, "tools", "micro", "dropin_time.cpp"),
                                   os.path.join(host, "tile_renderer_hip.cpp"), f"-L{pkg}", "-lmcrt", f"-Wl,-rpath,{pkg}", "-o", exe],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            p = subprocess.run([exe, str(cfg.width), str(cfg.height), str(cfg.maxBounces), str(cfg.samplesPerPixel), "12"], capture_output=True, text=True, timeout=300)
            return json.loads(p.stdout.strip().splitlines()[-1])
    except Exception as exc:
        return {"error": repr(exc)}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="1080p_b4_spp4_S64", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--quick-host", action="store_true", help="skip the cold-process call and the C++ drop-in binary (render_call keeps its medians)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the N>1 logic where RCCL cannot run (gathers through host memory)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--check", action="store_true", help="rank 0 also renders the whole frame alone and compares bit-for-bit")
    ap.add_argument("--frames-in-flight", type=int, default=4,
                    help="frames rendered concurrently (round-robin scene handles/streams/buffers); 1 = one frame at a time")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world

    import torch
    import torch.distributed as dist

    import minecraftskin_raytracer_amd as M
    from minecraftskin_raytracer_amd import abi

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the render path has no CPU fallback)")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    cfg, skin, pose = workload_config(M, args.workload)
    w, h, bounces, spp = cfg.width, cfg.height, cfg.maxBounces, cfg.samplesPerPixel
    sd = M.MeshBuilder.buildScene(M.synthetic_skin(skin), M.getBuiltinPoses()[pose])
    # F frames in flight: frame k is rendered by scene handle / stream / buffers k mod F.  One frame's
    # pipeline is a chain of 7 dependent launches, partly latency-bound; a renderer that produces a
    # sequence of frames overlaps the chain of one frame with the dense kernels of the next.  Every
    # frame is rendered completely; `latency_ms` below is the one-frame-at-a-time figure.
    F = max(1, args.frames_in_flight)
    # every handle plans its workspace against a budget of its own (default: a third of the HBM); F handles in
    # flight share the device, so each gets a share (a frame that needs more is cut into more passes)
    os.environ.setdefault("MCRT_WORKSPACE_MB", str(max(2048, 160 * 1024 // (F + 1))))
    scenes_ = [M.DeviceScene(sd, device=local_rank) for _ in range(F)]
    if F > 1:  # the frames in flight already fill the chip (and the hardware queues): one stream per frame
        for sc_ in scenes_:
            sc_.set_lanes(1)
    scene = scenes_[0]
    streams = [torch.cuda.Stream(device=dev) for _ in range(F)]
    frames = [torch.empty((h, w, 4), dtype=torch.float32, device=dev) for _ in range(F if world == 1 or rank == 0 else 0)]
    frame = frames[0] if frames else None

    if world == 1:
        def step(k: int) -> None:
            slot = k % F
            scenes_[slot].render_device(cfg, frames[slot].data_ptr(), 0, 1, abi.LAYOUT_FRAME, streams[slot].cuda_stream)

        def drain() -> None:
            pass
    else:
        from collections import deque
        tiles_y = (h + cfg.tileSize - 1) // cfg.tileSize
        max_rows = ((tiles_y + world - 1) // world) * cfg.tileSize  # padded so every rank sends the same count
        packed = [torch.empty((max_rows, w, 4), dtype=torch.float32, device=dev) for _ in range(F)]
        # rank 0: one contiguous [world, rows, W, 4] buffer per slot; the gather fills its per-rank views
        gathered = [torch.empty((world, max_rows, w, 4), dtype=torch.float32, device=dev) for _ in range(F)] if rank == 0 else []
        pending = deque()
        host_gather = args.dist_backend == "gloo"

        class _Done:
            def wait(self):
                return None

        def finish(k: int, slot: int, work) -> None:
            with torch.cuda.stream(streams[slot]):
                work.wait()
                if rank == 0:  # one launch un-permutes every rank's rows
                    M.assemble_frame_device(cfg, world, gathered[slot].data_ptr(), max_rows * w, frames[slot].data_ptr(),
                                            streams[slot].cuda_stream)

        def step(k: int) -> None:
            slot = k % F
            if len(pending) == F:  # step k - F used this slot: its gather must be done before the buffers are reused
                finish(*pending.popleft())
            with torch.cuda.stream(streams[slot]):
                buf = packed[slot]
                scenes_[slot].render_device(cfg, buf.data_ptr(), rank, world, abi.LAYOUT_PACKED, streams[slot].cuda_stream)
                if host_gather:  # rehearsal path: same sharding/assembly logic, collective through host memory
                    cpu = buf.cpu()
                    outs = [torch.empty_like(cpu) for _ in range(world)] if rank == 0 else None
                    dist.gather(cpu, outs, dst=0)
                    if rank == 0:
                        for r in range(world):
                            gathered[slot][r].copy_(outs[r])
                    work = _Done()
                else:
                    work = dist.gather(buf, list(gathered[slot].unbind(0)) if rank == 0 else None, dst=0, async_op=True)
            pending.append((k, slot, work))

        def drain() -> None:
            while pending:
                finish(*pending.popleft())

    def sync() -> None:
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # priming (not part of W or K): every slot's first renders allocate its workspace and record its
    # launch graph; one-time costs, like the extension build
    for k in range(5 * F):  # launch recording happens at a parameter set's 4th render
        step(k)
    drain()
    sync()
    for k in range(args.warmup):
        step(k)
    drain()
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    drain()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # one frame at a time (rank-local render only): enqueue, wait, repeat — on a handle of its own with the
    # library's defaults (a lone frame spreads itself over internal streams, mcrt_scene_set_lanes(0))
    lat_n = max(5, min(args.steps, 30))
    lat_target = frames[0] if world == 1 else packed[0]
    first, stepn = (0, 1) if world == 1 else (rank, world)
    layout = abi.LAYOUT_FRAME if world == 1 else abi.LAYOUT_PACKED
    stream = streams[0].cuda_stream
    lone = M.DeviceScene(sd, device=local_rank)
    for _ in range(6):  # workspace allocation, launch recording (4th render of a parameter set)
        lone.render_device(cfg, lat_target.data_ptr(), first, stepn, layout, stream)
    torch.cuda.synchronize()
    t_lat = time.perf_counter()
    for _ in range(lat_n):
        lone.render_device(cfg, lat_target.data_ptr(), first, stepn, layout, stream)
        torch.cuda.synchronize()
    latency_ms = (time.perf_counter() - t_lat) / lat_n * 1e3
    lone.check()
    lone.close()

    # device-only duration of one frame's pipeline on ONE stream: hipEvents on the launch stream
    target = frame if world == 1 else packed[0]
    pipeline_ms = scene.time_render_device(cfg, target.data_ptr(), max(5, min(args.steps, 50)), first, stepn, layout, stream)
    torch.cuda.synchronize()

    # wall time of one TileRenderer::render() call — host scene in, a fresh host Image out
    render_call = None
    if world == 1 and rank == 0:
        def timed(calls: int, **kw):
            walls, splits = [], []
            for i in range(2 + calls):
                t0 = time.perf_counter()
                M.TileRenderer.render(sd, cfg, **kw)
                dt = (time.perf_counter() - t0) * 1e3
                if M.TileRenderer.lastErrors():
                    raise SystemExit(f"render failed: {M.TileRenderer.lastErrors()}")
                if i >= 2:
                    walls.append(dt)
                    splits.append(M.TileRenderer.lastTimings())
            return walls, splits

        n_calls = max(10, min(args.steps, 20))
        walls, splits = timed(n_calls)  # out=None: a new Image(W, H) = (0,0,0,1) per call, inside the timed region
        host = np.zeros((h, w, 4), np.float32)
        reused, _ = timed(n_calls, out=host)
        untouched = []
        import mmap
        for _ in range(5):  # a destination whose pages were never touched (a fresh anonymous mapping): the runtime's pinning faults them in
            mm = mmap.mmap(-1, h * w * 16)
            fresh = np.frombuffer(mm, np.float32).reshape(h, w, 4)
            t0 = time.perf_counter()
            M.TileRenderer.render(sd, cfg, out=fresh)
            untouched.append((time.perf_counter() - t0) * 1e3)
            del fresh
            mm.close()
        med = statistics.median(walls)
        render_call = {
            "ms": round(med, 4),
            "mpixels_per_s": round(w * h / med / 1e3, 2),
            "calls": len(walls),
            "min_ms": round(min(walls), 4),
            "split_ms": {k: round(statistics.median(t[k] for t in splits), 4) for k in splits[0]},
            "reused_buffer_ms": round(statistics.median(reused), 4),
            "untouched_buffer_ms": round(statistics.median(untouched), 4),
            "what": "median wall time of TileRenderer.render (mcrt_render behind it) returning a FRESH Image per call, as the reference's call "
                    "site gets one: Image(W,H) allocation and (0,0,0,1) fill + scene flatten + upload + kernels + row-group downloads "
                    "overlapping the render + progress bookkeeping; reused_buffer_ms = the same call into one caller-owned buffer; "
                    "untouched_buffer_ms = into never-touched pages (a fresh anonymous mapping, what a bare malloc behind the C ABI is); split_ms = the "
                    "library's own split of the C call (mcrt_last_timings)",
        }
        if not args.quick_host:
            render_call["cold_process"] = cold_call(args.workload)
            render_call["cpp_dropin"] = cpp_dropin(cfg)

    check = None
    if args.check and rank == 0:
        whole = torch.empty_like(frame)
        scene.render_device(cfg, whole.data_ptr(), 0, 1, abi.LAYOUT_FRAME, stream)
        torch.cuda.synchronize()
        check = all(bool(torch.equal(whole, f)) for f in frames[:min(F, args.steps + args.warmup)])

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mpix = w * h * args.steps / elapsed / 1e6
        owned_rows = scene.owned_pixel_rows(cfg, first, stepn)
        owned_px = min(owned_rows, h) * w if world > 1 else w * h
        algo_bytes = 16.0 * owned_px  # 16 B written per output pixel, ~0 read
        achieved = algo_bytes / (pipeline_ms * 1e-3) / 1e9
        # counters of one frame of this workload from the committed rocprofv3 --pmc passes (tools/pmc_run.sh)
        pmc = {}
        prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(prof):
            try:
                pmc = json.load(open(prof)).get(args.workload) or {}
                if not isinstance(pmc, dict):  # older files held the traffic alone
                    pmc = {"hbm_bytes": pmc}
            except Exception:
                pmc = {}
        traffic = pmc.get("hbm_bytes")
        valu = pmc.get("valu_wave_instructions")
        src_hash = kernel_source_hash()
        pmc_stale = bool(pmc) and pmc.get("source_hash") != src_hash  # the counters were taken on other kernel sources
        roof = {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "kernel": "one frame's whole pipeline on one stream (kernel.pipeline_ms; the per-kernel split is in " + str(pmc.get("profile", "profiles/")) + ")",
            "algorithmic_bytes": algo_bytes,
            "counters": {"from": pmc.get("profile"), "source_hash": pmc.get("source_hash"), "this_build": src_hash, "stale": pmc_stale} if pmc else None,
            "note": "algorithmic bytes = 16 B x output pixels of the frame; the path is VALU/latency-bound by construction (DESIGN.md), "
                    "so the HBM fraction is << 1 %: the VALU figures are the ones that describe kernel quality",
        }
        # frames that share the device are launched in other shapes (fewer workgroups, one wave per tile stream): their own
        # counters when the PMC passes took them (tools/collect_profiles.sh), else the lone frame's
        piped = pmc.get("pipelined") if isinstance(pmc.get("pipelined"), dict) and pmc["pipelined"].get("source_hash") == pmc.get("source_hash") else None
        traffic_p = (piped or {}).get("hbm_bytes", traffic)
        valu_p = (piped or {}).get("valu_wave_instructions", valu)
        if traffic and world == 1:
            roof["traffic_over_algorithmic"] = round(traffic / algo_bytes, 2)
            roof["hbm_traffic_gbs_pipelined"] = round(traffic_p / (ms_per_step * 1e-3) / 1e9, 1)
        if valu and world == 1:
            lane_ops = valu * 64.0  # a wave-instruction = 64 lane-ops
            roof["valu_wave_instructions"] = valu
            roof["valu_peak_tlaneops"] = VALU_PEAK_TLOPS
            roof["valu_frac"] = round(lane_ops / (pipeline_ms * 1e-3) / (VALU_PEAK_TLOPS * 1e12), 4)  # one frame at a time
            roof["valu_frac_pipelined"] = round(valu_p * 64.0 / (ms_per_step * 1e-3) / (VALU_PEAK_TLOPS * 1e12), 4)  # at `value`
            if piped:
                roof["pipelined_counters"] = {"traffic": traffic_p, "valu_wave_instructions": valu_p,
                                              "note": "one frame launched as `value`'s frames are (MCRT_SHARED_GRIDS=1)"}
        line = {
            "metric": "Mpixels/s",
            "value": round(mpix, 2),
            "unit": "Mpixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{w}x{h}, {bounces} bounces, {spp} spp, 1 light, synthetic 64x{64 if skin == 'S64' else 32} skin ({skin}), pose {pose}, soft shadows 8, tile 32"
                            + (", AO 16, DOF aperture 0.3 (reference GUI defaults)" if args.workload == "gui_defaults" else ""),
                "name": args.workload,
                "value_is": f"device throughput, {F} frame(s) in flight, scene and frame resident in HBM (the bench contract's `value`: inputs "
                            "resident, PCIe excluded); latency_ms = one frame at a time on the device; render_call.ms = SURVEY 8(d)'s t, one "
                            "host-to-host TileRenderer::render call returning a fresh Image (the PCIe-inclusive figure, never `value`)",
                "parallelism": "single GPU" if world == 1 else f"cyclic tile rows over {world} GPUs + RCCL gather to rank 0 (overlapped)",
                "frames_in_flight": F,
                "hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
            },
            "latency_ms": round(latency_ms, 4),
            "latency_mpixels_per_s": round(w * h / latency_ms / 1e3, 2),
            "kernel": {"pipeline_ms": round(pipeline_ms, 4),
                       "note": "hipEvents on the launch stream around one frame's whole pipeline on one stream (plan_tiles, "
                               "primary, [ao,] lit, resolve)"},
            "roofline": roof,
        }
        if render_call is not None:
            line["render_call"] = render_call
        if check is not None:
            line["check_assembled_frame_equals_single_gpu_render"] = check
        if world == 1 and not args.no_cpu_baseline:
            try:
                line["cpu_baseline"] = cpu_baseline(args.workload, gpu_frame=frames[0].cpu().numpy())
            except Exception as exc:  # the baseline is a report, never a reason to lose the GPU number
                line["cpu_baseline"] = {"error": repr(exc)}
        print(json.dumps(line), flush=True)

    for sc_ in scenes_:
        sc_.check()  # raises if the device flagged an internal inconsistency during any render
        sc_.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
