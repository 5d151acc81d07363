"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs.  Bar: bit-exact float32 (and therefore bit-exact RGBA8)."""
import numpy as np
import pytest

import scenes
from minecraftskin_raytracer_amd import abi

pytestmark = pytest.mark.gpu


def test_library_loaded_is_in_tree(mcrt, gpu):
    from minecraftskin_raytracer_amd import _lib

    assert _lib.LIB_PATH.endswith("minecraftskin_raytracer_amd/libmcrt.so")
    assert mcrt.device_count() >= 1


# ---- device arithmetic vs the host build of the same header --------------------------------------
def test_detmath_device_matches_host_on_render_domain(mcrt, gpu):
    two_pi_bits = int(np.float32(6.2831855).view(np.uint32))
    assert mcrt.probe_detmath_range(0, 0, two_pi_bits + 16) == 0  # sinf on [0, 2pi]
    assert mcrt.probe_detmath_range(1, 0, two_pi_bits + 16) == 0  # cosf on [0, 2pi]
    # the fused sincos the kernels use for light / lens / AO samples, against the separate host functions
    assert mcrt.probe_detmath_range(3, 0, two_pi_bits + 16) == 0
    assert mcrt.probe_detmath_range(4, 0, two_pi_bits + 16) == 0
    neg0 = 0x80000000
    assert mcrt.probe_detmath_range(3, neg0, neg0 + two_pi_bits + 16) == 0  # negative arguments
    assert mcrt.probe_detmath_range(4, neg0, neg0 + two_pi_bits + 16) == 0
    one_bits = int(np.float32(1.0).view(np.uint32))
    assert mcrt.probe_detmath_range(2, 0, one_bits + 64, 16.0) == 0  # powf(x,16) on [0, 1+]


def test_fast_reciprocal_is_the_ieee_reciprocal(mcrt, gpu):
    """rt::rcp_exact (hardware rcp + one FMA Newton step, used for the per-ray reciprocals) against the
    host's IEEE 1.0f / x for EVERY float with 2^-126 <= |x| <= 2^126, both signs."""
    f = lambda v: int(np.float32(v).view(np.uint32))
    lo, hi = f(2.0 ** -126), f(2.0 ** 126)
    assert mcrt.probe_detmath_range(5, lo, hi) == 0
    assert mcrt.probe_detmath_range(5, lo | 0x80000000, hi | 0x80000000) == 0


def test_frame_division_is_the_ieee_division(mcrt, gpu):
    """rt::div_frame ((px + jitter) / width as a product with 1/width and one FMA correction) against the general
    division for EVERY float a sample coordinate can take (0 and 2^-33 .. d + 1): the frame sizes of BASELINE.json, the
    ends of the verified range and a block of 300 odd sizes; tools/gpu_verify_div.py covers every divisor up to 16384
    (profiles/r02_v5/div_const_exhaustive.txt).  The probe itself is checked with the uncorrected product."""
    from minecraftskin_raytracer_amd import api

    for d in (1, 2, 3, 256, 1080, 1920, 2160, 3840, 4320, 7680, 16383, 16384):
        assert api.probe_div_const(d, 1) == (0, 0), d
    assert api.probe_div_const(601, 300) == (0, 0)
    assert api.probe_div_const(1920, 1, mode=2)[0] > 1000000  # the probe's own check: the uncorrected product does differ
    # rt::sqrt_pos (square roots without the general expansion's rescaling of tiny arguments): 0 and every float from 2^-96 to infinity
    assert api.probe_div_const(1, 1, mode=3) == (0, 0)
    # the reciprocals come from the host (RenderParams::inv_width / inv_height, and the probe's too): the device's own
    # 1.0f / d agrees with them for every divisor of the verified range
    for first in range(1, 16385, 4096):
        assert api.probe_div_const(first, 4096, mode=4) == (0, 0), first


def test_detmath_device_random(mcrt, gpu):
    g = np.random.default_rng(1)
    x = g.uniform(-200, 200, 1 << 16).astype(np.float32)
    import ctypes as C

    # host values through the oracle-free path: probe on device vs numpy float64 sanity (1 ulp) and
    # exactness is covered by the range test; here just check large arguments don't blow up
    s = mcrt.probe_detmath(0, x)
    c = mcrt.probe_detmath(1, x)
    assert np.max(np.abs(s - np.sin(x.astype(np.float64)))) < 1e-6
    assert np.max(np.abs(c - np.cos(x.astype(np.float64)))) < 1e-6


def test_mt19937_uniform_device(mcrt, gpu, oracle):
    seeds = [0, 1, 5489, 12345, 0xFFFFFFFF, 0x80000000, 2463534242]
    for n in (16, 128, 227):
        dev = mcrt.probe_mt_uniform(seeds, n)
        for i, s in enumerate(seeds):
            scenes.assert_bit_equal(dev[i], oracle.mt_uniform(s, n), f"mt seed {s} n {n}")
    dev = mcrt.probe_mt_uniform(seeds[:3], 1500)  # long-stream engine
    for i, s in enumerate(seeds[:3]):
        scenes.assert_bit_equal(dev[i], oracle.mt_uniform(s, 1500), f"mt long seed {s}")


# ---- intersectScene ------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,pose", [("S64", 0), ("S64", 6), ("S64", 3), ("S32", 1)])
def test_intersect_scene_matches_oracle(mcrt, gpu, oracle, kind, pose):
    sd = scenes.skin_scene(kind, pose)
    ds = mcrt.DeviceScene(sd)
    rays = scenes.random_rays(6000, seed=pose + 11)
    scenes.assert_hits_equal(ds.intersect(rays), oracle.intersect(sd.ptr, rays), f"{kind} pose {pose}")
    ds.close()


def test_intersect_reference_unit_cases(mcrt, gpu, oracle):
    # test_intersection.cpp:20-66: front hit t=4, +X side, miss, behind
    red = scenes.solid((1, 0, 0, 1))
    sc = scenes.simple_scene([scenes.build_box(red, (0, 0, 0), (2, 2, 2))])
    sd = mcrt.SceneDesc(sc)
    ds = mcrt.DeviceScene(sd)
    rays = np.array([[0, 0, 5, 0, 0, -1], [5, 0, 0, -1, 0, 0], [0, 5, 5, 0, 0, -1], [0, 0, 5, 0, 0, 1]], np.float32)
    h = ds.intersect(rays)
    assert list(h["hit"]) == [1, 1, 0, 0]
    assert abs(h["t"][0] - 4.0) < 1e-4 and abs(h["point"][0][2] - 1.0) < 1e-4 and abs(h["normal"][0][2] - 1.0) < 1e-4
    assert abs(h["t"][1] - 4.0) < 1e-4 and abs(h["normal"][1][0] - 1.0) < 1e-4
    scenes.assert_hits_equal(h, oracle.intersect(sd.ptr, rays))
    ds.close()


# ---- traceRay --------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfgkw", [dict(maxBounces=0), dict(maxBounces=4), dict(maxBounces=3, softShadows=False),
                                   dict(maxBounces=2, aoEnabled=True, aoSamples=16), dict(maxBounces=1, gradientBg=False)])
def test_trace_matches_oracle(mcrt, gpu, oracle, cfgkw):
    sd = scenes.skin_scene("S64", 6)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(**cfgkw)
    rays = scenes.random_rays(3000, seed=5)
    scenes.assert_bit_equal(ds.trace(cfg, rays, 0), oracle.trace(sd.ptr, cfg, rays, 0, cfg.maxBounces), str(cfgkw))
    ds.close()


# ---- full renders -----------------------------------------------------------------------------------
RENDER_CASES = [
    ("S64", 0, dict(width=96, height=54, maxBounces=1, samplesPerPixel=1)),
    ("S64", 0, dict(width=96, height=54, maxBounces=4, samplesPerPixel=4)),
    ("S64", 6, dict(width=96, height=54, maxBounces=4, samplesPerPixel=4)),
    ("S64", 3, dict(width=64, height=64, maxBounces=2, samplesPerPixel=2, tileSize=7)),
    ("S64", 0, dict(width=64, height=64, maxBounces=2, samplesPerPixel=3, aoEnabled=True, dofEnabled=True)),
    ("S32", 1, dict(width=80, height=45, maxBounces=8, samplesPerPixel=16, tileSize=16)),
    ("S64", 5, dict(width=70, height=50, maxBounces=3, samplesPerPixel=5, softShadows=False, gradientBg=False)),
    ("S64", 0, dict(width=33, height=31, maxBounces=0, samplesPerPixel=1, tileSize=64)),
    ("S64", 2, dict(width=40, height=40, maxBounces=2, samplesPerPixel=300, tileSize=20, shadowSamples=2)),
    ("S64", 0, dict(width=48, height=48, maxBounces=-1, samplesPerPixel=1)),
]


@pytest.mark.parametrize("kind,pose,cfgkw", RENDER_CASES)
def test_render_matches_oracle(mcrt, gpu, oracle, kind, pose, cfgkw):
    sd = scenes.skin_scene(kind, pose)
    cfg = abi.Config(**cfgkw)
    img = mcrt.TileRenderer.render(sd, cfg)
    assert mcrt.TileRenderer.lastErrors() == []
    ref = oracle.render(sd.ptr, cfg)
    scenes.assert_bit_equal(img, ref, f"{kind} pose {pose} {cfgkw}")
    assert np.array_equal(mcrt.quantize_rgba8(img), oracle.quantize(ref).reshape(img.shape))


def test_render_default_scene_and_empty_scene(mcrt, gpu, oracle):
    cfg = abi.Config(width=64, height=64, maxBounces=1)
    sd = mcrt.MeshBuilder.buildDefaultScene()
    scenes.assert_bit_equal(mcrt.TileRenderer.render(sd, cfg), oracle.render(sd.ptr, cfg), "default scene")
    empty = mcrt.SceneDesc(scenes.simple_scene())
    scenes.assert_bit_equal(mcrt.TileRenderer.render(empty, cfg), oracle.render(empty.ptr, cfg), "empty scene")


def test_progress_callback_contract(mcrt, gpu):
    # test_tile_renderer.cpp:85-104: exactly totalTiles calls, total constant
    calls = []
    cfg = abi.Config(width=32, height=32, maxBounces=0, tileSize=16)
    mcrt.TileRenderer.render(mcrt.SceneDesc(scenes.simple_scene()), cfg, lambda d, t: calls.append((d, t)))
    assert [c[0] for c in calls] == [1, 2, 3, 4] and all(c[1] == 4 for c in calls)


# ---- rarely taken kernel variants ----------------------------------------------------------------------
VARIANT_CASES = [
    # non-power-of-two shadow sample counts → lit counts through atomics instead of a wave ballot
    ("S64", 6, dict(width=72, height=48, maxBounces=2, samplesPerPixel=2, shadowSamples=5)),
    ("S64", 0, dict(width=72, height=48, maxBounces=3, samplesPerPixel=1, shadowSamples=13)),
    # power-of-two counts other than 8, including a whole wave per hit
    ("S64", 3, dict(width=64, height=40, maxBounces=1, samplesPerPixel=1, shadowSamples=64)),
    ("S64", 0, dict(width=64, height=40, maxBounces=2, samplesPerPixel=2, shadowSamples=16)),
    # per-hit RNG streams longer than 227 draws → "general" kernels with the full mt19937 engine
    ("S64", 0, dict(width=48, height=32, maxBounces=1, samplesPerPixel=1, shadowSamples=120)),
    ("S64", 5, dict(width=40, height=30, maxBounces=1, samplesPerPixel=1, aoEnabled=True, aoSamples=130)),
    # deep recursion (stack stride > 16), hard shadows, point light via radius handled below
    ("S64", 0, dict(width=48, height=48, maxBounces=20, samplesPerPixel=1, softShadows=False)),
    # huge tile (one tile = whole frame, many parts) and tile size 1 (one pixel per tile)
    ("S64", 6, dict(width=70, height=44, maxBounces=2, samplesPerPixel=3, tileSize=128)),
    ("S32", 1, dict(width=24, height=20, maxBounces=1, samplesPerPixel=2, tileSize=1)),
    # DOF with many samples (4 draws per sample), large spp → sample-per-thread background path
    ("S64", 0, dict(width=40, height=24, maxBounces=1, samplesPerPixel=40, dofEnabled=True, aperture=0.8, focusDistance=45.0)),
    # ambient occlusion as its own stage (ballot counts for power-of-two sample counts, atomics otherwise),
    # with and without soft shadows, more AO directions than light samples and fewer
    ("S64", 0, dict(width=96, height=64, maxBounces=2, samplesPerPixel=2, aoEnabled=True, aoSamples=16)),
    ("S64", 6, dict(width=80, height=60, maxBounces=1, samplesPerPixel=1, aoEnabled=True, aoSamples=5, aoRadius=6.0, aoIntensity=0.9)),
    ("S64", 3, dict(width=64, height=48, maxBounces=3, samplesPerPixel=2, aoEnabled=True, aoSamples=4, shadowSamples=16)),
    ("S32", 0, dict(width=64, height=48, maxBounces=1, samplesPerPixel=1, aoEnabled=True, aoSamples=64, softShadows=False)),
    ("S64", 0, dict(width=48, height=32, maxBounces=0, samplesPerPixel=3, aoEnabled=True, aoSamples=1)),
    # the reference GUI's default feature set (AO 16 + DOF), small
    ("S64", 0, dict(width=120, height=68, maxBounces=4, samplesPerPixel=8, aoEnabled=True, aoSamples=16, dofEnabled=True, aperture=0.3)),
    # DOF with tile culling: small tiles, blur circles from a few pixels to wider than the character
    # (focus in front of / on / far behind the figure, auto focus), posed and un-posed
    ("S64", 0, dict(width=240, height=160, maxBounces=1, samplesPerPixel=2, tileSize=8, dofEnabled=True, aperture=0.3)),
    ("S64", 6, dict(width=240, height=160, maxBounces=1, samplesPerPixel=2, tileSize=8, dofEnabled=True, aperture=2.5, focusDistance=20.0)),
    ("S64", 3, dict(width=200, height=120, maxBounces=0, samplesPerPixel=3, tileSize=5, dofEnabled=True, aperture=1.2, focusDistance=400.0)),
    ("S32", 0, dict(width=160, height=200, maxBounces=1, samplesPerPixel=2, tileSize=16, dofEnabled=True, aperture=6.0, focusDistance=50.0)),
]


@pytest.mark.parametrize("kind,pose,cfgkw", VARIANT_CASES)
def test_render_variants_match_oracle(mcrt, gpu, oracle, kind, pose, cfgkw):
    sd = scenes.skin_scene(kind, pose)
    cfg = abi.Config(**cfgkw)
    img = mcrt.TileRenderer.render(sd, cfg)
    assert mcrt.TileRenderer.lastErrors() == []
    scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), f"{kind} pose {pose} {cfgkw}")


def test_point_light_soft_shadow_degenerates(mcrt, gpu, oracle):
    # light.radius < 1e-4 with softShadows on: computeSoftShadow falls back to one hard ray, raw normal
    d = scenes.skin_scene("S64", 2).to_numpy()
    sc = abi.scene_from_numpy(d)
    sc.light_radius = 0.0
    sd = mcrt.SceneDesc(sc)
    cfg = abi.Config(width=64, height=48, maxBounces=2, samplesPerPixel=2)
    scenes.assert_bit_equal(mcrt.TileRenderer.render(sd, cfg), oracle.render(sd.ptr, cfg), "point light")


def test_scene_too_big_for_lds_tables(mcrt, gpu, oracle):
    # > 64 meshes: the face/mesh tables do not fit the LDS budget → kernels compiled against HBM views,
    # and meshes beyond bit 63 of the candidate mask take the tail loops
    g = np.random.default_rng(7)
    meshes = []
    for i in range(70):
        col = tuple(g.uniform(0.2, 1.0, 3)) + (1.0,)
        pos = (float(g.uniform(-12, 12)), float(g.uniform(4, 30)), float(g.uniform(-6, 6)))
        tex = abi.Texture(2, 2, np.array([col, col, col, (0, 0, 0, 0) if i % 3 == 0 else col], np.float32))
        meshes.append(scenes.build_box(tex, pos, (2.0, 2.0, 2.0), 0.5 if i % 3 == 0 else 0.0))
    sc = scenes.simple_scene(meshes, light=(0, 40, 30), cam_pos=(0, 18, 50), cam_target=(0, 18, 0), bg=(0.2, 0.3, 0.5, 1.0))
    sd = mcrt.SceneDesc(sc)
    cfg = abi.Config(width=96, height=64, maxBounces=3, samplesPerPixel=2)
    scenes.assert_bit_equal(mcrt.TileRenderer.render(sd, cfg), oracle.render(sd.ptr, cfg), "70-mesh scene")
    ds = mcrt.DeviceScene(sd)
    rays = scenes.random_rays(2000, seed=3, target=(0.0, 18.0, 0.0), spread=14.0)
    scenes.assert_hits_equal(ds.intersect(rays), oracle.intersect(sd.ptr, rays), "70-mesh intersect")
    ds.close()


def test_big_texture_pool_stays_in_hbm(mcrt, gpu, oracle):
    # > 64 Ki texels: alpha predicates cannot be staged in LDS
    g = np.random.default_rng(11)
    px = g.uniform(0, 1, size=(300 * 300, 4)).astype(np.float32)
    px[g.uniform(size=len(px)) < 0.3, 3] = 0.0
    tex = abi.Texture(300, 300, px)
    inner = scenes.build_box(scenes.solid((0.9, 0.8, 0.7, 1.0)), (0, 18, 0), (8, 8, 8))
    outer = scenes.build_box(tex, (0, 18, 0), (8, 8, 8), 0.5)
    sd = mcrt.SceneDesc(scenes.simple_scene([inner, outer], light=(0, 40, 30), cam_pos=(0, 18, 40), bg=(0.2, 0.3, 0.5, 1.0)))
    cfg = abi.Config(width=80, height=60, maxBounces=2, samplesPerPixel=2)
    scenes.assert_bit_equal(mcrt.TileRenderer.render(sd, cfg), oracle.render(sd.ptr, cfg), "big texture")


@pytest.mark.parametrize("spp,dof,tile", [(33, False, 32), (37, True, 32), (64, False, 24), (64, True, 32), (70, False, 50), (48, True, 7)])
def test_high_sample_counts_take_the_slab_background_kernel(mcrt, oracle, gpu, spp, dof, tile):
    """From 33 samples per pixel the background tiles are rendered by `background_kernel`: a wave stages slabs of 64 pixels x
    16 samples of the tile's stream in LDS (coalesced), a lane per pixel consumes them in sample order.  Sample counts that are
    no multiple of the slab, clipped and odd-sized tiles, depth of field (the stream then holds the jitter pairs only)."""
    import scenes

    sd = scenes.skin_scene("S64", 3)
    kw = dict(width=150, height=100, maxBounces=2, samplesPerPixel=spp, tileSize=tile)
    if dof:
        kw.update(dofEnabled=True, aperture=0.3)
    cfg = mcrt.Config(**kw)
    img = mcrt.TileRenderer.render(sd, cfg)
    assert mcrt.TileRenderer.lastErrors() == []
    scenes.assert_bit_equal(img, oracle.render(sd.ptr, cfg), f"spp {spp} dof {dof} tile {tile}")


@pytest.mark.parametrize("bounces,spp,size,tile", [(4, 2, (96, 64), 32), (8, 1, (70, 50), 16), (1, 3, (64, 48), 32), (3, 4, (50, 37), 7),
                                                    (3, 2, (1920, 1080), 32)])  # the last: 16 000 blocks of level-1 records, more than any grid
def test_closed_room_fills_every_chain_to_the_last_level(mcrt, oracle, gpu, bounces, spp, size, tile):
    """The worst case the record queues are laid out for: the camera stands inside a closed room (one large box) with two
    boxes in it, so every sample hits, every reflection hits again, and every chain runs to maxBounces.  `lit` follows the
    chains of a block of 256 level-1 records into a region of 256 x (maxBounces - 1) slots that the block owns: here every
    region is filled to its last slot (frame and tile sizes that leave partly filled blocks and units included)."""
    wall = scenes.solid((0.7, 0.75, 0.8, 1.0))
    room = scenes.build_box(wall, (0, 18, 0), (60, 40, 60))
    a = scenes.build_box(scenes.solid((0.9, 0.2, 0.2, 1.0)), (-5, 8, -6), (8, 16, 8))
    b = scenes.build_box(scenes.solid((0.2, 0.8, 0.3, 1.0)), (7, 5, 2), (6, 10, 6))
    sc = scenes.simple_scene([room, a, b], light=(3, 34, 5), cam_pos=(2, 16, 24), cam_target=(0, 12, 0), radius=2.0)
    sd = mcrt.SceneDesc(sc)
    cfg = abi.Config(width=size[0], height=size[1], maxBounces=bounces, samplesPerPixel=spp, tileSize=tile)
    want = oracle.render(sd.ptr, cfg)
    scenes.assert_bit_equal(mcrt.TileRenderer.render(sd, cfg), want, f"closed room, {bounces} bounces")
    ds = mcrt.DeviceScene(sd)
    import torch

    out = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    for lanes in (1, 3):  # and through the device entry point, alone and with lanes
        ds.set_lanes(lanes)
        out.zero_()
        ds.render_device(cfg, out.data_ptr(), 0, 1, abi.LAYOUT_FRAME, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        scenes.assert_bit_equal(out.cpu().numpy(), want, f"closed room, {lanes} lane(s)")
    ds.check()
    ds.close()
