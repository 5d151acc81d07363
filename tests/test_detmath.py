"""include/mcrt_detmath.h against the system libm (sampled; `tools/check_detmath 1` is the
exhaustive run: 0 mismatches over all 2^32 floats for sinf/cosf and over x in [0,2] for
powf(x,16) on glibc 2.35's FMA variants, the ones every FMA-capable x86-64 host selects)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_has_fma() -> bool:
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return False
    return " fma " in flags and " avx2 " in flags


def test_detmath_matches_glibc_fma_variants(tmp_path):
    check_against_host_libm(tmp_path)


@pytest.mark.gpu
def test_detmath_matches_the_gpu_box_hosts_glibc(tmp_path):
    """The same check in the `-m gpu` run: it needs no GPU, but the host whose libm the compiled reference (bench.py's
    cpu_baseline, oracle/_ref) uses there is the GPU box's, not the dev container's."""
    check_against_host_libm(tmp_path)


def check_against_host_libm(tmp_path):
    if not cpu_has_fma():
        pytest.skip("host CPU lacks FMA/AVX2: glibc picks its SSE2 sinf/cosf/powf bodies, which differ in ~1e-8 of inputs")
    exe = tmp_path / "check_detmath"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", f"-I{ROOT}/include",
                           f"{ROOT}/tools/check_detmath.cpp", "-o", str(exe), "-lpthread", "-lm"])
    out = subprocess.run([str(exe), "509"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "sin_mismatch=0 cos_mismatch=0 sincos_mismatch=0 pow16_mismatch=0 pow_random_mismatch=0" in out.stdout
