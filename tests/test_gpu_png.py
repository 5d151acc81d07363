"""RGBA8 fused into the render epilogue and the one-call render → PNG path."""
import numpy as np
import pytest

import scenes
from minecraftskin_raytracer_amd import abi
from test_png import decode_png

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cfgkw", [
    dict(width=200, height=120, maxBounces=2, samplesPerPixel=2),
    dict(width=97, height=61, maxBounces=1, samplesPerPixel=1, tileSize=7),
    dict(width=128, height=96, maxBounces=1, samplesPerPixel=20, dofEnabled=True, aperture=0.5),  # sample-per-thread background path
])
def test_fused_rgba8_equals_quantised_float_frame(mcrt, gpu, oracle, cfgkw):
    import torch

    cfg = abi.Config(**cfgkw)
    sd = scenes.skin_scene("S64", 6)
    ds = mcrt.DeviceScene(sd)
    st = torch.cuda.current_stream().cuda_stream
    f32 = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    both_f = torch.zeros_like(f32)
    both_q = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.uint8, device="cuda")
    only_q = torch.zeros_like(both_q)
    ds.render_device(cfg, f32.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
    ds.render_device_ex(cfg, both_f.data_ptr(), both_q.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
    ds.render_device_ex(cfg, 0, only_q.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
    # a rank shard, packed, RGBA8 only
    rows = ds.owned_pixel_rows(cfg, 1, 2)
    packed_q = torch.zeros((rows, cfg.width, 4), dtype=torch.uint8, device="cuda")
    ds.render_device_ex(cfg, 0, packed_q.data_ptr(), 1, 2, abi.LAYOUT_PACKED, st)
    torch.cuda.synchronize()
    ref = oracle.render(sd.ptr, cfg)
    scenes.assert_bit_equal(f32.cpu().numpy(), ref, "float frame vs oracle")
    assert torch.equal(both_f, f32)
    want = mcrt.quantize_rgba8(ref)  # the reference quantiser on the reference-equal frame
    assert np.array_equal(both_q.cpu().numpy(), want)
    assert np.array_equal(only_q.cpu().numpy(), want)
    T = cfg.tileSize
    got = packed_q.cpu().numpy()
    k = 0
    for r in range(1, (cfg.height + T - 1) // T, 2):
        y0, y1 = r * T, min(cfg.height, (r + 1) * T)
        assert np.array_equal(got[k * T:k * T + (y1 - y0)], want[y0:y1]), r
        k += 1
    ds.close()


def test_render_png_matches_render_then_quantise(mcrt, gpu, tmp_path):
    cfg = abi.Config(width=160, height=90, maxBounces=3, samplesPerPixel=4)
    sd = scenes.skin_scene("S64", 3)
    path = str(tmp_path / "frame.png")
    assert mcrt.render_png(sd, cfg, path)
    img = mcrt.TileRenderer.render(sd, cfg)
    assert np.array_equal(decode_png(open(path, "rb").read()), mcrt.quantize_rgba8(img))
    assert not mcrt.render_png(sd, abi.Config(width=0, height=10), str(tmp_path / "none.png"))
