"""Randomised HIP-vs-oracle parity (bit-exact float frames), fixed seeds; the long sweep is tools/gpu_fuzz.py."""
import pytest

import scenes
from fuzz_cases import make_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("block", range(4))
def test_random_cases_match_oracle(mcrt, gpu, oracle, block):
    for seed in range(1000 + 12 * block, 1000 + 12 * (block + 1)):
        sd, cfg, what = make_case(seed)
        img = mcrt.TileRenderer.render(sd, cfg)
        assert mcrt.TileRenderer.lastErrors() == [], what
        scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), what)
