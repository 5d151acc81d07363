"""Compiles the C++ drop-in TileRenderer (csrc/host/tile_renderer_hip.cpp) + the restated
reference TileRenderer tests (tests/cpp/test_drop_in.cpp) with g++ and runs them: against the
mirror types everywhere, and against the reference's OWN headers where /root/reference exists
(source compatibility of the drop-in with the reference tree)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "minecraftskin_raytracer_amd", "csrc", "host")
PKG = os.path.join(ROOT, "minecraftskin_raytracer_amd")
REF = "/root/reference/src"


def build(tmp_path, with_reference: bool) -> str:
    exe = str(tmp_path / ("drop_in_ref" if with_reference else "drop_in"))
    cmd = ["g++", "-std=c++17", "-O1", f"-I{ROOT}/include", f"-I{HOST}", os.path.join(ROOT, "tests", "cpp", "test_drop_in.cpp"),
           os.path.join(HOST, "tile_renderer_hip.cpp"), os.path.join(HOST, "image_writer_hip.cpp"), f"-L{PKG}", "-lmcrt", f"-Wl,-rpath,{PKG}", "-o", exe]
    if with_reference:
        cmd[3:3] = ["-DMCRT_USE_REFERENCE_HEADERS", f"-I{REF}"]
    subprocess.check_call(cmd)
    return exe


def run(exe, *args, env=None):
    p = subprocess.run([exe, *args], capture_output=True, text=True, env=dict(os.environ, **(env or {})))
    assert p.returncode == 0, p.stdout + p.stderr
    return p.stdout


def test_drop_in_host_behaviour_with_mirror_types(mcrt, tmp_path):
    if mcrt.device_count() > 0:
        pytest.skip("host-only variant is for boxes without a GPU")
    out = run(build(tmp_path, False))
    assert "host: 0 failure(s)" in out and "no HIP device" in out


def test_drop_in_compiles_against_reference_headers(mcrt, tmp_path):
    if not os.path.exists(os.path.join(REF, "raytracer", "tile_renderer.h")):
        pytest.skip("/root/reference is not present here")
    exe = build(tmp_path, True)
    if mcrt.device_count() == 0:
        assert "host: 0 failure(s)" in run(exe)


@pytest.mark.gpu
def test_drop_in_renders_on_gpu(mcrt, gpu, tmp_path):
    assert "gpu: 0 failure(s)" in run(build(tmp_path, False), "--gpu")


@pytest.mark.gpu
@pytest.mark.parametrize("gather", ["0", "1"])
def test_drop_in_renders_on_all_devices(mcrt, gpu, tmp_path, gather):
    """MCRT_DEVICE=all routes the same TileRenderer::render call through mcrt_render_multi (every visible
    device takes its cyclic share of the tile rows); the restated reference tests must still hold."""
    assert "gpu: 0 failure(s)" in run(build(tmp_path, False), "--gpu", env={"MCRT_DEVICE": "all", "MCRT_GATHER": gather})
