"""Full-size checks at BASELINE.json's configurations, through properties that do not need a full
CPU render: tile independence (any sharding reassembles to the same frame, which is also the
multi-GPU contract), idempotence, oracle renderTile on a sample of tiles of the full-size frame
(incl. the heaviest ones), culling on/off equivalence via a DOF-disabled/enabled-aperture-0 pair,
packed layout + unpack kernel, device quantiser."""
import numpy as np
import pytest
import torch

import scenes
from minecraftskin_raytracer_amd import abi

pytestmark = pytest.mark.gpu


def render_dev(mcrt, ds, cfg, first=0, step=1, layout=abi.LAYOUT_FRAME, rows=None):
    h = rows if rows is not None else cfg.height
    out = torch.zeros((h, cfg.width, 4), dtype=torch.float32, device="cuda")
    ds.render_device(cfg, out.data_ptr(), first, step, layout, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


@pytest.fixture(scope="module")
def north_star(mcrt, gpu):
    sd = scenes.skin_scene("S64", 0)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=4)
    frame = render_dev(mcrt, ds, cfg)
    yield sd, ds, cfg, frame
    ds.close()


def test_1080p_idempotent_and_sane(mcrt, north_star):
    sd, ds, cfg, frame = north_star
    again = render_dev(mcrt, ds, cfg)
    assert torch.equal(frame, again)
    f = frame.cpu().numpy()
    assert np.isfinite(f).all() and f.min() >= 0.0 and f.max() <= 1.0 and (f[..., 3] == 1.0).all()
    assert (np.abs(f[..., :3] - f[0, 0, :3]).sum(axis=2) > 0.05).mean() > 0.03  # the character is there


@pytest.mark.parametrize("world", [2, 8])
def test_1080p_tile_row_shards_reassemble(mcrt, north_star, world):
    sd, ds, cfg, frame = north_star
    from minecraftskin_raytracer_amd import parallel

    rebuilt = torch.zeros_like(frame)
    rows = parallel.packed_rows(cfg, world)
    for r in range(world):
        assert ds.owned_pixel_rows(cfg, r, world) == len(parallel.owned_tile_rows(cfg, r, world)) * cfg.tileSize
        packed = render_dev(mcrt, ds, cfg, r, world, abi.LAYOUT_PACKED, rows)
        mcrt.unpack_rows_device(cfg, r, world, packed.data_ptr(), rebuilt.data_ptr(), torch.cuda.current_stream().cuda_stream)
        # the CPU un-permute used by the gloo tests agrees with the HIP unpack kernel
        cpu = torch.zeros((cfg.height, cfg.width, 4))
        parallel.unpack_rows(cfg, r, world, packed.cpu(), cpu)
        own = torch.zeros(cfg.height, dtype=torch.bool)
        for tr in parallel.owned_tile_rows(cfg, r, world):
            own[tr * cfg.tileSize:(tr + 1) * cfg.tileSize] = True
        torch.cuda.synchronize()
        assert torch.equal(cpu[own], frame.cpu()[own])
    torch.cuda.synchronize()
    assert torch.equal(rebuilt, frame)


def test_1080p_sampled_tiles_match_oracle(mcrt, oracle, north_star):
    sd, ds, cfg, frame = north_star
    f = frame.cpu().numpy()
    tiles = oracle.generate_tiles(cfg.width, cfg.height, cfg.tileSize)
    # heaviest tiles = most non-background pixels, plus corners, clipped bottom row, random ones
    bgdiff = np.abs(f[..., :3] - f[0, 0, :3]).sum(axis=2) > 0.2
    weight = [bgdiff[y:y + h, x:x + w].sum() for x, y, w, h in tiles]
    pick = list(np.argsort(weight)[-6:]) + [0, 59, len(tiles) - 1, len(tiles) - 60] + list(np.random.default_rng(0).integers(0, len(tiles), 6))
    scratch = np.zeros_like(f)
    for i in pick:
        oracle.render_tile(sd.ptr, cfg, tiles[i], scratch)
        x, y, w, h = tiles[i]
        scenes.assert_bit_equal(f[y:y + h, x:x + w], scratch[y:y + h, x:x + w], f"tile {i} {tiles[i]}")


@pytest.mark.parametrize("scale,spp", [(1.0, 4), (0.8, 4), (1.7, 1), (1.0, 13), (0.0, 2)])
def test_one_colour_background_tiles_match_oracle(mcrt, oracle, north_star, scale, spp):
    """Background tiles where the gradient's `dist` clamps to 1 for every sample are filled with the edge colour without
    draws (render_kernels.hip: constant_background): the tiles on both sides of that criterion — the ring where the
    gradient reaches the edge colour — against the oracle's renderTile, for several radii and sample counts (13 spp:
    the background goes through `primary`)."""
    sd, ds, _, _ = north_star
    cfg = abi.Config(width=1920, height=1080, maxBounces=4, samplesPerPixel=spp, gradientScale=scale)
    f = render_dev(mcrt, ds, cfg).cpu().numpy()
    tiles = oracle.generate_tiles(cfg.width, cfg.height, cfg.tileSize)

    def reach(t):  # the smallest gradient distance over the tile's rectangle
        x, y, w, h = t
        ulo, uhi, vlo, vhi = x / cfg.width, (x + w) / cfg.width, y / cfg.height, (y + h) / cfg.height
        cx = 0.0 if ulo <= 0.5 <= uhi else min(abs(ulo - 0.5), abs(uhi - 0.5))
        cy = 0.0 if vlo <= 0.5 <= vhi else min(abs(vlo - 0.5), abs(vhi - 0.5))
        return 2.0 * scale * (cx * cx + cy * cy) ** 0.5

    ring = [i for i, t in enumerate(tiles) if 0.93 <= reach(t) <= 1.07]
    g = np.random.default_rng(int(scale * 100) + spp)
    pick = list(g.choice(ring, size=min(24, len(ring)), replace=False)) if ring else []
    pick += [0, 59, len(tiles) - 1, len(tiles) - 60, 30]  # corners, top middle
    scratch = np.zeros_like(f)
    for i in pick:
        oracle.render_tile(sd.ptr, cfg, tiles[i], scratch)
        x, y, w, h = tiles[i]
        scenes.assert_bit_equal(f[y:y + h, x:x + w], scratch[y:y + h, x:x + w], f"scale {scale} spp {spp} tile {i} {tiles[i]}")


def test_4k_b8_spp16_sampled_tiles_match_oracle(mcrt, oracle, gpu):
    # BASELINE.json configs[2]
    sd = scenes.skin_scene("S64", 0)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=3840, height=2160, maxBounces=8, samplesPerPixel=16)
    f = render_dev(mcrt, ds, cfg).cpu().numpy()
    ds.close()
    tiles = oracle.generate_tiles(cfg.width, cfg.height, cfg.tileSize)
    bgdiff = np.abs(f[..., :3] - f[0, 0, :3]).sum(axis=2) > 0.2
    weight = [bgdiff[y:y + h, x:x + w].sum() for x, y, w, h in tiles]
    scratch = np.zeros_like(f)
    for i in list(np.argsort(weight)[-2:]) + [0, len(tiles) - 1, 3000]:
        oracle.render_tile(sd.ptr, cfg, tiles[i], scratch)
        x, y, w, h = tiles[i]
        scenes.assert_bit_equal(f[y:y + h, x:x + w], scratch[y:y + h, x:x + w], f"tile {i}")


def test_4k_b4_spp4_sampled_tiles_and_8_way_shards(mcrt, oracle, gpu):
    # BASELINE.json configs[3]: 3840x2160, 4 bounces, 4 spp, tile rows sharded 8 ways (68 tile rows, the last
    # clipped to 16 px) — at full size: sampled tiles against the oracle's renderTile, and the 8 ranks' packed
    # shards re-assembled by the gather root's launch equal the whole-frame render
    sd = scenes.skin_scene("S64", 0)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=3840, height=2160, maxBounces=4, samplesPerPixel=4)
    frame = render_dev(mcrt, ds, cfg)
    f = frame.cpu().numpy()
    tiles = oracle.generate_tiles(cfg.width, cfg.height, cfg.tileSize)
    assert len(tiles) == 120 * 68 and tiles[-1][3] == 16
    bgdiff = np.abs(f[..., :3] - f[0, 0, :3]).sum(axis=2) > 0.2
    weight = [bgdiff[y:y + h, x:x + w].sum() for x, y, w, h in tiles]
    scratch = np.zeros_like(f)
    for i in list(np.argsort(weight)[-4:]) + [0, 119, len(tiles) - 1, len(tiles) - 120, 4000]:
        oracle.render_tile(sd.ptr, cfg, tiles[i], scratch)
        x, y, w, h = tiles[i]
        scenes.assert_bit_equal(f[y:y + h, x:x + w], scratch[y:y + h, x:x + w], f"tile {i} {tiles[i]}")
    from minecraftskin_raytracer_amd import parallel

    world = 8
    rows = parallel.packed_rows(cfg, world)
    st = torch.cuda.current_stream().cuda_stream
    gathered = torch.full((world, rows, cfg.width, 4), -3.0, dtype=torch.float32, device="cuda")
    for r in range(world):
        ds.render_device(cfg, gathered[r].data_ptr(), r, world, abi.LAYOUT_PACKED, st)
    rebuilt = torch.zeros_like(frame)
    mcrt.assemble_frame_device(cfg, world, gathered.data_ptr(), rows * cfg.width, rebuilt.data_ptr(), st)
    torch.cuda.synchronize()
    ds.check()
    assert torch.equal(rebuilt, frame)
    ds.close()


def test_8k_legacy_skin_band_matches_oracle(mcrt, oracle, gpu):
    # BASELINE.json configs[4] geometry (7680x4320, 8 bounces, 64 spp, legacy skin, single reference
    # light) on one owned tile row of an 8-way shard; oracle on two tiles of it
    sd = scenes.skin_scene("S32", 0)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=7680, height=4320, maxBounces=8, samplesPerPixel=64)
    tiles_y = (cfg.height + 31) // 32
    row = 62  # a tile row through the character, owned by rank 62 % 8 = 6 of 8
    out = torch.zeros((32, cfg.width, 4), dtype=torch.float32, device="cuda")
    ds.render_device(cfg, out.data_ptr(), row, tiles_y, abi.LAYOUT_PACKED, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    ds.close()
    f = out.cpu().numpy()
    scratch = np.zeros((cfg.height, cfg.width, 4), np.float32)
    for tx in (120, 0):
        tile = (tx * 32, row * 32, 32, 32)
        oracle.render_tile(sd.ptr, cfg, tile, scratch)
        scenes.assert_bit_equal(f[:, tx * 32:tx * 32 + 32], scratch[row * 32:row * 32 + 32, tx * 32:tx * 32 + 32], f"8k tile {tile}")


def test_8k_whole_frame_idempotent_shards_and_oracle_tiles(mcrt, oracle, gpu):
    """BASELINE.json configs[4] at FULL size: 7680x4320, 8 bounces, 64 spp, legacy 64x32 skin, the single reference
    light (the reference's Scene holds one Light).  The whole frame: idempotent, equal to its 8 ranks' packed shards
    re-assembled by the gather root's launch, and equal to the oracle's renderTile on eight tiles — the four that
    hold most of the figure, the frame's corners and two background tiles (tile_renderer.cpp:71-127)."""
    sd = scenes.skin_scene("S32", 0)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=7680, height=4320, maxBounces=8, samplesPerPixel=64)
    frame = render_dev(mcrt, ds, cfg)
    again = render_dev(mcrt, ds, cfg)
    assert torch.equal(frame, again)
    del again
    from minecraftskin_raytracer_amd import parallel

    world = 8
    rows = parallel.packed_rows(cfg, world)
    st = torch.cuda.current_stream().cuda_stream
    gathered = torch.full((world, rows, cfg.width, 4), -3.0, dtype=torch.float32, device="cuda")
    for r in range(world):
        ds.render_device(cfg, gathered[r].data_ptr(), r, world, abi.LAYOUT_PACKED, st)
    rebuilt = torch.zeros_like(frame)
    mcrt.assemble_frame_device(cfg, world, gathered.data_ptr(), rows * cfg.width, rebuilt.data_ptr(), st)
    torch.cuda.synchronize()
    ds.check()
    assert torch.equal(rebuilt, frame)
    del rebuilt, gathered
    f = frame.cpu().numpy()
    ds.close()
    assert np.isfinite(f).all() and f.min() >= 0.0 and f.max() <= 1.0 and (f[..., 3] == 1.0).all()
    tiles = oracle.generate_tiles(cfg.width, cfg.height, cfg.tileSize)
    assert len(tiles) == 240 * 135
    bgdiff = np.abs(f[..., :3] - f[0, 0, :3]).sum(axis=2) > 0.2
    weight = [bgdiff[y:y + h, x:x + w].sum() for x, y, w, h in tiles]
    scratch = np.zeros_like(f)
    for i in list(np.argsort(weight)[-4:]) + [0, len(tiles) - 1, 239, 240 * 67 + 5]:
        oracle.render_tile(sd.ptr, cfg, tiles[i], scratch)
        x, y, w, h = tiles[i]
        scenes.assert_bit_equal(f[y:y + h, x:x + w], scratch[y:y + h, x:x + w], f"8k tile {i} {tiles[i]}")


def test_culling_never_changes_the_image(mcrt, gpu):
    # dofEnabled with aperture below the 1e-6 gate renders the pinhole path; enabling DOF with a tiny
    # aperture above the gate disables primary-ray culling.  A direct A/B of the culled kernel: render
    # with a camera that makes culling impossible (fov 179.9 → bounds exceed every tile) is not
    # equivalent, so compare against the probe path, which never culls.
    sd = scenes.skin_scene("S64", 6)
    ds = mcrt.DeviceScene(sd)
    cfg = abi.Config(width=160, height=120, maxBounces=1, samplesPerPixel=1)
    img = render_dev(mcrt, ds, cfg).cpu().numpy()
    # primary rays through pixel centres, traced by the un-culled probe
    import ctypes as C
    ys, xs = np.mgrid[0:cfg.height, 0:cfg.width]
    u = ((xs.astype(np.float32) + np.float32(0.5)) / np.float32(cfg.width)).ravel()
    v = ((ys.astype(np.float32) + np.float32(0.5)) / np.float32(cfg.height)).ravel()
    import oraclelib
    orc = oraclelib.Oracle()
    rays = np.stack([orc.camera_ray(sd.ptr, float(a), float(b), cfg.width / cfg.height) for a, b in zip(u[::7], v[::7])])
    hits = ds.intersect(rays)
    traced = ds.trace(cfg, rays, 0)
    flat = img.reshape(-1, 4)[::7]
    m = hits["hit"] != 0
    scenes.assert_bit_equal(flat[m], traced[m], "hit pixels: culled kernel vs un-culled probe")
    ds.close()


def test_device_quantiser_matches_host(mcrt, north_star):
    sd, ds, cfg, frame = north_star
    q = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.uint8, device="cuda")
    mcrt.quantize_rgba8_device(frame.data_ptr(), q.data_ptr(), cfg.width * cfg.height, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(q.cpu().numpy(), mcrt.quantize_rgba8(frame.cpu().numpy()))


def test_multi_batch_render_matches_single_batch(mcrt, gpu, tmp_path):
    """A tiny workspace budget forces the wavefront pipeline to cut the frame into many batches of
    tile rows; the image must not change.  (The budget is read once per process → subprocess.)"""
    import os
    import subprocess
    import sys

    script = tmp_path / "render_small_budget.py"
    script.write_text(
        "import sys, numpy as np\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.abspath(__file__)))})\n"
        "import minecraftskin_raytracer_amd as M, scenes\n"
        "cfg = M.Config(width=320, height=200, maxBounces=3, samplesPerPixel=4)\n"
        "img = M.TileRenderer.render(scenes.skin_scene('S64', 6), cfg)\n"
        "assert M.TileRenderer.lastErrors() == []\n"
        "np.save(sys.argv[1], img)\n")
    out_small, out_big = str(tmp_path / "small.npy"), str(tmp_path / "big.npy")
    env = dict(os.environ, MCRT_WORKSPACE_MB="8")  # ~2 tile rows per batch at this size
    subprocess.check_call([sys.executable, str(script), out_small], env=env)
    subprocess.check_call([sys.executable, str(script), out_big])
    a, b = np.load(out_small), np.load(out_big)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    import oraclelib
    cfg = abi.Config(width=320, height=200, maxBounces=3, samplesPerPixel=4)
    sd = scenes.skin_scene("S64", 6)  # keep the description alive while the oracle reads it
    ref = oraclelib.Oracle().render(sd.ptr, cfg)
    scenes.assert_bit_equal(a, ref, "multi-batch render vs oracle")


def test_lanes_render_the_same_frame(mcrt, gpu, tmp_path):
    """A render is split over 1..4 lanes (streams with their own workspace, every n-th tile row of the
    shard each); whole frames and packed rank shards must come out bit-identical for any lane
    count, also combined with a tiny budget.  (MCRT_LANES is read once per process → subprocess.)"""
    import os
    import subprocess
    import sys

    script = tmp_path / "render_lanes.py"
    script.write_text(
        "import sys, numpy as np, torch\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))})\n"
        f"sys.path.insert(0, {repr(os.path.dirname(os.path.abspath(__file__)))})\n"
        "import minecraftskin_raytracer_amd as M, scenes\n"
        "from minecraftskin_raytracer_amd import abi\n"
        "cfg = M.Config(width=333, height=250, maxBounces=3, samplesPerPixel=4, tileSize=16)\n"
        "ds = M.DeviceScene(scenes.skin_scene('S64', 6))\n"
        "st = torch.cuda.current_stream().cuda_stream\n"
        "frame = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device='cuda')\n"
        "ds.render_device(cfg, frame.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)\n"
        "rows = ds.owned_pixel_rows(cfg, 1, 3)\n"
        "packed = torch.zeros((rows, cfg.width, 4), dtype=torch.float32, device='cuda')\n"
        "ds.render_device(cfg, packed.data_ptr(), 1, 3, abi.LAYOUT_PACKED, st)\n"
        "shard = torch.zeros_like(frame)\n"
        "M.unpack_rows_device(cfg, 1, 3, packed.data_ptr(), shard.data_ptr(), st)\n"
        "torch.cuda.synchronize()\n"
        "np.savez(sys.argv[1], frame=frame.cpu().numpy(), shard=shard.cpu().numpy())\n")
    outs = {}
    for lanes, budget in (("1", None), ("2", None), ("3", None), ("4", None), ("3", "8")):
        env = dict(os.environ, MCRT_LANES=lanes)
        if budget:
            env["MCRT_WORKSPACE_MB"] = budget
        path = str(tmp_path / f"lanes{lanes}_{budget}.npz")
        subprocess.check_call([sys.executable, str(script), path], env=env)
        outs[(lanes, budget)] = np.load(path)
    base = outs[("1", None)]
    T, H = 16, 250
    owned = np.zeros(H, bool)
    for r in range(1, (H + T - 1) // T, 3):
        owned[r * T:min(H, (r + 1) * T)] = True
    assert np.array_equal(base["shard"][owned].view(np.uint32), base["frame"][owned].view(np.uint32))
    assert not base["shard"][~owned].any()
    for key, o in outs.items():
        assert np.array_equal(o["frame"].view(np.uint32), base["frame"].view(np.uint32)), key
        assert np.array_equal(o["shard"].view(np.uint32), base["shard"].view(np.uint32)), key
    import oraclelib
    cfg = abi.Config(width=333, height=250, maxBounces=3, samplesPerPixel=4, tileSize=16)
    sd = scenes.skin_scene("S64", 6)
    scenes.assert_bit_equal(base["frame"], oraclelib.Oracle().render(sd.ptr, cfg), "lanes render vs oracle")


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_assemble_frame_equals_per_rank_unpack(mcrt, gpu, world):
    """The gather root's single-launch assembly of all ranks' packed rows gives the frame that the
    per-rank un-permute gives, and both equal the whole-frame render."""
    import torch

    cfg = abi.Config(width=250, height=170, maxBounces=2, samplesPerPixel=2, tileSize=16)
    ds = mcrt.DeviceScene(scenes.skin_scene("S64", 3))
    st = torch.cuda.current_stream().cuda_stream
    whole = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    ds.render_device(cfg, whole.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
    tiles_y = (cfg.height + cfg.tileSize - 1) // cfg.tileSize
    max_rows = ((tiles_y + world - 1) // world) * cfg.tileSize
    gathered = torch.full((world, max_rows, cfg.width, 4), -7.0, dtype=torch.float32, device="cuda")
    for r in range(world):
        ds.render_device(cfg, gathered[r].data_ptr(), r, world, abi.LAYOUT_PACKED, st)
    a = torch.zeros_like(whole)
    b = torch.zeros_like(whole)
    mcrt.assemble_frame_device(cfg, world, gathered.data_ptr(), max_rows * cfg.width, a.data_ptr(), st)
    for r in range(world):
        mcrt.unpack_rows_device(cfg, r, world, gathered[r].data_ptr(), b.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(a, whole) and torch.equal(b, whole)
    ds.close()


def test_render_survives_a_nearly_full_device(mcrt, gpu):
    """With most of HBM taken by someone else the workspace budget is halved until the buffers fit;
    the frame is the same (smaller batches only)."""
    import torch

    cfg = abi.Config(width=1280, height=720, maxBounces=3, samplesPerPixel=4)
    sd = scenes.skin_scene("S64", 0)
    st = torch.cuda.current_stream().cuda_stream
    ref = torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda")
    ds = mcrt.DeviceScene(sd)
    ds.render_device(cfg, ref.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
    torch.cuda.synchronize()
    ds.close()
    mcrt.trim()  # the closed scene's workspace would otherwise be handed to the next scene as is
    free, _total = torch.cuda.mem_get_info()
    leave = 100 << 20  # the unconstrained workspace of this frame is ~0.25 GB (sized for the tiles meshes can touch)
    hog = torch.empty(max(0, free - leave), dtype=torch.uint8, device="cuda")
    try:
        free2, _ = torch.cuda.mem_get_info()
        assert free2 < (400 << 20)
        ds2 = mcrt.DeviceScene(sd)
        # the output frame is carved out of the hog so that it does not compete for the leftover
        frame = hog[: cfg.height * cfg.width * 16].view(torch.float32).view(cfg.height, cfg.width, 4)
        ds2.render_device(cfg, frame.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st)
        torch.cuda.synchronize()
        ds2.check()
        assert torch.equal(frame, ref)
        ds2.close()
        mcrt.trim()
    finally:
        frame = None
        del hog
        torch.cuda.empty_cache()


def test_recorded_launches_replay_the_same_frame(mcrt, gpu, oracle):
    """A render's launches are issued directly the first time a parameter set is seen, recorded as a
    graph the second time and replayed afterwards; more parameter sets than record slots evict each
    other.  Every repetition must give the same bits."""
    import torch

    sd = scenes.skin_scene("S64", 6)
    ds = mcrt.DeviceScene(sd)
    st = torch.cuda.Stream()
    cfgs = [abi.Config(width=96 + 8 * i, height=64, maxBounces=2, samplesPerPixel=2, tileSize=16) for i in range(6)]
    want = [oracle.render(sd.ptr, c) for c in cfgs]
    outs = [torch.zeros((c.height, c.width, 4), dtype=torch.float32, device="cuda") for c in cfgs]
    for rep in range(4):  # 6 parameter sets cycle through 4 slots and keep evicting each other
        for c, o in zip(cfgs, outs):
            o.zero_()
            ds.render_device(c, o.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
        torch.cuda.synchronize()
        for c, o, w in zip(cfgs, outs, want):
            scenes.assert_bit_equal(o.cpu().numpy(), w, f"cycle {rep}, width {c.width}")
    for rep in range(5):  # two sets repeated 5 times: direct x3, recorded at the 4th sighting, replayed
        a = torch.zeros_like(outs[0])
        ds.render_device(cfgs[0], a.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
        ds.render_device(cfgs[0], outs[0].data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
        torch.cuda.synchronize()
        scenes.assert_bit_equal(a.cpu().numpy(), want[0], f"repeat {rep} (fresh buffer)")
        scenes.assert_bit_equal(outs[0].cpu().numpy(), want[0], f"repeat {rep}")
    ds.set_lanes(3)  # forced lanes go through the same recording
    for rep in range(3):
        outs[1].zero_()
        ds.render_device(cfgs[1], outs[1].data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
        torch.cuda.synchronize()
        scenes.assert_bit_equal(outs[1].cpu().numpy(), want[1], f"3 lanes, repeat {rep}")
    ds.close()


def test_workspace_is_reused_across_scenes_and_trim_frees_it(mcrt, gpu, oracle):
    """A destroyed scene leaves its workspace for the next scene on the device (one-shot renders would
    otherwise re-allocate it every call); results do not depend on whose workspace is used; trim() frees it."""
    import torch

    mcrt.trim()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    cfg = abi.Config(width=1280, height=720, maxBounces=2, samplesPerPixel=4)
    a = scenes.skin_scene("S64", 0)
    img_a = mcrt.TileRenderer.render(a, cfg)  # creates, renders, destroys → workspace pooled
    free1, _ = torch.cuda.mem_get_info()
    assert free1 < free0 - (30 << 20)  # something is being kept
    b = scenes.skin_scene("S32", 6)  # different scene (posed, other tables) on the same shell
    img_b = mcrt.TileRenderer.render(b, cfg)
    free2, _ = torch.cuda.mem_get_info()
    assert abs(free2 - free1) < (64 << 20)  # no second workspace
    scenes.assert_bit_equal(img_a, oracle.render(a.ptr, cfg), "first scene")
    scenes.assert_bit_equal(img_b, oracle.render(b.ptr, cfg), "second scene on the pooled workspace")
    mcrt.trim()
    free3, _ = torch.cuda.mem_get_info()
    assert free3 > free1 + (30 << 20)


def test_frames_in_flight_on_one_device_render_the_same_bits(mcrt, gpu, oracle):
    """A render enqueued while another handle's frame is still running on the device sizes its launches for sharing
    (fewer workgroups per kernel, api.cpp device_shared / choose_grids), one enqueued on an idle device for its own
    latency; both are recorded as launch graphs of their own.  The schedule must not show in the pixels."""
    sd = scenes.skin_scene("S64", 6)
    cfg = abi.Config(width=480, height=272, maxBounces=3, samplesPerPixel=2)
    want = oracle.render(sd.ptr, cfg)
    handles = [mcrt.DeviceScene(sd) for _ in range(4)]
    streams = [torch.cuda.Stream() for _ in handles]
    outs = [torch.zeros((cfg.height, cfg.width, 4), dtype=torch.float32, device="cuda") for _ in handles]
    for rep in range(8):  # in flight together: all but the very first enqueue find the device busy
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        for ds, st, o in zip(handles, streams, outs):
            ds.render_device(cfg, o.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
        torch.cuda.synchronize()
        for i, o in enumerate(outs):
            scenes.assert_bit_equal(o.cpu().numpy(), want, f"round {rep}, handle {i} (frames in flight)")
    for rep in range(6):  # one at a time: idle device
        for i, (ds, st, o) in enumerate(zip(handles, streams, outs)):
            o.zero_()
            ds.render_device(cfg, o.data_ptr(), 0, 1, abi.LAYOUT_FRAME, st.cuda_stream)
            torch.cuda.synchronize()
            scenes.assert_bit_equal(o.cpu().numpy(), want, f"round {rep}, handle {i} (alone)")
    for ds in handles:
        ds.check()
        ds.close()
