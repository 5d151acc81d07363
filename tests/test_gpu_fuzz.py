"""Randomised HIP-vs-oracle parity (bit-exact float frames), fixed seeds; the long sweep is tools/gpu_fuzz.py."""
import pytest

import scenes
from fuzz_cases import make_bundle_case, make_case

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
