"""The HIP path against the committed fixtures generated from the compiled reference
(tests/golden/, tools/make_golden.py) — no oracle in the loop."""
import json
import os

import numpy as np
import pytest

import scenes
from minecraftskin_raytracer_amd import abi

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RENDERS = json.load(open(os.path.join(GOLDEN, "renders.json")))


@pytest.mark.parametrize("case", RENDERS, ids=[c["name"] for c in RENDERS])
def test_render_equals_reference_fixture(mcrt, gpu, case):
    g = np.load(os.path.join(GOLDEN, f"render_{case['name']}.npz"))
    sd = (mcrt.MeshBuilder.buildDefaultScene(mcrt.getBuiltinPoses()[case["pose"]]) if case["skin"] == "default"
          else scenes.skin_scene(case["skin"], case["pose"]))
    img = mcrt.TileRenderer.render(sd, abi.Config(**case["config"]))
    assert mcrt.TileRenderer.lastErrors() == []
    scenes.assert_bit_equal(img, g["image"], case["name"])        # float colour: identical, not just within 1e-5
    assert np.array_equal(mcrt.quantize_rgba8(img), g["rgba8"])   # integer RGBA after the reference quantisation


@pytest.mark.parametrize("name", ["S64_pose0", "S64_pose6", "S32_pose1"])
def test_probes_equal_reference_vectors(mcrt, gpu, name):
    g = np.load(os.path.join(GOLDEN, f"vectors_{name}.npz"))
    sd = scenes.skin_scene(str(g["skin"]), int(g["pose"]))
    ds = mcrt.DeviceScene(sd)
    scenes.assert_hits_equal(ds.intersect(g["rays"]), g["hits"], "intersectScene")
    scenes.assert_bit_equal(ds.trace(abi.Config(maxBounces=2), g["rays"][:400], 0), g["trace"], "traceRay")
    ds.close()


def test_mt19937_equals_libstdcxx_fixture(mcrt, gpu):
    g = np.load(os.path.join(GOLDEN, "rng.npz"))
    dev = mcrt.probe_mt_uniform([int(s) for s in g["seeds"]], 128)
    scenes.assert_bit_equal(dev, g["draws"], "mt19937 + uniform_real_distribution<float>")
