"""bench.py end to end on the GPU box: the single-GPU line and a 2-rank rehearsal of the N > 1 control
flow (tile-row shards, packed buffers, gather, single-launch assembly, frames in flight).  The
rehearsal puts both ranks on the one GPU and gathers through gloo/host memory — RCCL needs one
device per rank — so it checks logic and results, not speed."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _last_json(text: str) -> dict:
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError(text[-2000:])


def test_bench_single_gpu_line(gpu):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--check"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _last_json(p.stdout)
    assert d["metric"] == "Mpixels/s" and d["unit"] == "Mpixels/s" and d["n_gpus"] == 1 and d["steps"] == 12 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["check_assembled_frame_equals_single_gpu_render"] is True
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and r["traffic"] > 0
    assert d["value"] > 100 and abs(d["value"] - 1920 * 1080 / (d["ms_per_step"] * 1e-3) / 1e6) / d["value"] < 0.01


def test_bench_two_rank_rehearsal(gpu):
    import socket

    with socket.socket() as sock:  # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "9", "--warmup", "2",
           "--dist-backend", "gloo", "--single-device", "--check"]
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    d = _last_json(p.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "cpu_baseline" not in d
    assert d["check_assembled_frame_equals_single_gpu_render"] is True
