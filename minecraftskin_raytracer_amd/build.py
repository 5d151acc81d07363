"""Builds the product library ``libmcrt.so`` (HIP kernels for gfx950 + C-ABI host code) in-tree.

    python -m minecraftskin_raytracer_amd.build [--force] [--verbose]

hipcc cross-compiles for gfx950 without a GPU.  Flags that matter for parity:
``-ffp-contract=off`` (no fused multiply-add except the explicit fma() calls of mcrt_detmath.h) and
hipcc's default correctly-rounded fp32 divide/sqrt.  No fast-math.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libmcrt.so")
SOURCES = ["render_kernels.hip", "api.cpp", "flatten.cpp", "scene_builder.cpp", "png_writer.cpp"]
HEADERS = ["flat_scene.h", "flatten.h", "kernels.h", "rt_core.h"]
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or add /opt/rocm/bin to PATH)")


def _stale() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS]
    deps += [os.path.join(ROOT, "include", f) for f in ("mcrt.h", "mcrt_detmath.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return OUT
    cmd = [
        _hipcc(),
        f"--offload-arch={ARCH}",
        "-O3",
        "-std=c++17",
        "-ffp-contract=off",
        "-fPIC",
        "-shared",
        "-Wall",
        "-Wno-unused-function",
        f"-I{os.path.join(ROOT, 'include')}",
        f"-I{CSRC}",
    ]
    if verbose:
        cmd += ["-Rpass-analysis=kernel-resource-usage"]
    cmd += os.environ.get("MCRT_EXTRA_FLAGS", "").split()
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-o", OUT, "-lpthread"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
