"""Randomised HIP-vs-oracle parity (bit-exact float frames), fixed seeds; the long sweep is tools/gpu_fuzz.py."""
import pytest

import scenes
from fuzz_cases import make_bundle_case, make_case, make_wide_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_random_cases_match_oracle(mcrt, gpu, oracle, block):
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        sd, cfg, what = make_case(seed)
        img = mcrt.TileRenderer.render(sd, cfg)
        assert mcrt.TileRenderer.lastErrors() == [], what
        scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), what)


@pytest.mark.parametrize("block", range(4))
def test_bundle_decision_cases_match_oracle(mcrt, gpu, oracle, block):
    """Scenes aimed at `lit`'s whole-bundle shadow decisions (lights near, inside and grazing boxes; texel grids of
    every density; scaled scenes): a hit declared all-lit or all-shadowed without tracing must agree with the oracle."""
    for seed in range(7000 + 15 * block, 7000 + 15 * (block + 1)):
        sd, cfg, what = make_bundle_case(seed)
        img = mcrt.TileRenderer.render(sd, cfg)
        assert mcrt.TileRenderer.lastErrors() == [], what
        scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), what)


@pytest.mark.parametrize("block", range(4))
def test_scaled_and_far_away_scenes_match_oracle(mcrt, gpu, oracle, block):
    """The same scene types scaled by 1e-5 ... 1e6 or moved up to 3e6 away from the origin (one ulp of a coordinate up to
    0.25), lights a thousand times smaller than the scene: the margins of the conservative masks, bounds and decisions are
    multiples of the scene's coordinate magnitude (flat_scene.h: mask_slack), so they hold at any scale."""
    for seed in range(9000 + 15 * block, 9000 + 15 * (block + 1)):
        sd, cfg, what = make_wide_case(seed)
        img = mcrt.TileRenderer.render(sd, cfg)
        assert mcrt.TileRenderer.lastErrors() == [], what
        scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), what)
