"""The pure-host parts of the library (scene builder, flattener, PNG encoder + quantiser) under AddressSanitizer + UBSan.
GPU sanitizers are not available on the pool; the device side is covered by the parity suites."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "minecraftskin_raytracer_amd", "csrc")


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not available")
def test_host_only_code_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "asan_host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-ffp-contract=off", f"-I{ROOT}/include", f"-I{CSRC}",
                           os.path.join(ROOT, "tests", "cpp", "asan_host.cpp"), os.path.join(CSRC, "flatten.cpp"),
                           os.path.join(CSRC, "scene_builder.cpp"), os.path.join(CSRC, "png_writer.cpp"), "-o", exe])
    p = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert p.returncode == 0, p.stdout + p.stderr
    assert "asan driver: 0 problem(s)" in p.stdout and "ERROR" not in p.stderr and "runtime error" not in p.stderr
