"""Host-side checks that need no GPU: the C-ABI library loads and exports every symbol the header
declares, and the host logic (tile grid, config defaults, quantiser, flattener, scene builder)
behaves like the reference."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import scenes
from minecraftskin_raytracer_amd import abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol(mcrt):
    from minecraftskin_raytracer_amd import _lib

    header = open(os.path.join(ROOT, "include", "mcrt.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mcrt_[a-z0-9_]+)\s*\(", header)) - {"mcrt_progress_fn"})
    assert len(declared) >= 20
    lib = _lib.load()
    for name in declared:
        assert hasattr(lib, name), f"libmcrt.so does not export {name}"
    assert sorted(_lib.EXPORTED_SYMBOLS) == declared
    assert lib.mcrt_abi_version() == 3


def test_struct_layouts_match_header_sizes():
    assert C.sizeof(abi.McrtConfig) == 17 * 4 + 8 * 4
    assert C.sizeof(abi.McrtHit) == 52
    assert C.sizeof(abi.McrtTile) == 16


def test_config_defaults(mcrt):
    from minecraftskin_raytracer_amd import _lib

    c = abi.McrtConfig()
    _lib.load().mcrt_config_init(C.byref(c))
    d = abi.Config().to_c()
    assert bytes(c) == bytes(d)  # raytracer.h:10-38 defaults on both sides
    assert (c.width, c.height, c.max_bounces, c.samples_per_pixel, c.tile_size, c.shadow_samples) == (256, 256, 3, 1, 32, 8)


# ---- TileRenderer::generateTiles — /root/reference/tests/test_tile_renderer.cpp:9-57 ------------
def test_generate_tiles_exact_division(mcrt):
    t = mcrt.TileRenderer.generateTiles(64, 64, 32)
    assert t == [(0, 0, 32, 32), (32, 0, 32, 32), (0, 32, 32, 32), (32, 32, 32, 32)]


def test_generate_tiles_with_remainder(mcrt):
    assert mcrt.TileRenderer.generateTiles(50, 30, 32) == [(0, 0, 32, 30), (32, 0, 18, 30)]


def test_generate_tiles_small_image(mcrt):
    assert mcrt.TileRenderer.generateTiles(10, 10, 32) == [(0, 0, 10, 10)]


def test_generate_tiles_invalid_input(mcrt):
    for args in ((0, 64, 32), (64, 0, 32), (64, 64, 0), (-1, 64, 32)):
        assert mcrt.TileRenderer.generateTiles(*args) == []


def test_generate_tiles_cover_without_overlap(mcrt, oracle):
    # test_tile_renderer_props.cpp:30-80: coverage, no overlap, grid alignment
    g = np.random.default_rng(0)
    for _ in range(40):
        w, h, ts = int(g.integers(1, 700)), int(g.integers(1, 700)), int(g.integers(1, 257))
        tiles = mcrt.TileRenderer.generateTiles(w, h, ts)
        assert tiles == oracle.generate_tiles(w, h, ts)
        cover = np.zeros((h, w), np.int32)
        for x, y, tw, th in tiles:
            assert x % ts == 0 and y % ts == 0 and 0 < tw <= ts and 0 < th <= ts
            cover[y:y + th, x:x + tw] += 1
        assert (cover == 1).all()


# ---- quantiser — /root/reference/tests/test_image_writer.cpp:55-118 -----------------------------
def test_quantize_known_answers(mcrt, oracle):
    img = np.array([[[1, 0, 0, 1], [0, 0, 0, 0.5], [2.0, -0.5, 0.25, 1.0]]], np.float32)
    q = mcrt.quantize_rgba8(img)
    assert q[0, 0].tolist() == [255, 0, 0, 255]
    assert q[0, 1, 3] == 128
    assert q[0, 2].tolist() == [255, 0, 64, 255]
    g = np.random.default_rng(2)
    r = g.uniform(-0.2, 1.2, size=(64, 64, 4)).astype(np.float32)
    assert np.array_equal(mcrt.quantize_rgba8(r), oracle.quantize(r).reshape(r.shape))


# ---- render without a GPU must fail loudly, never fall back -------------------------------------
def test_render_without_device_records_error(mcrt):
    if mcrt.device_count() > 0:
        pytest.skip("a GPU is visible here")
    img = mcrt.TileRenderer.render(mcrt.MeshBuilder.buildDefaultScene(), abi.Config(width=8, height=8))
    errs = mcrt.TileRenderer.lastErrors()
    assert len(errs) == 1 and errs[0][0] == -1 and "no HIP device" in errs[0][1]
    assert img.shape == (8, 8, 4) and (img[..., :3] == 0).all() and (img[..., 3] == 1).all()
    with pytest.raises(Exception):
        mcrt.DeviceScene(mcrt.MeshBuilder.buildDefaultScene())


def test_invalid_sizes_return_untouched_image(mcrt):
    # tile_renderer.cpp:144-146: zero tiles → Image(w,h) returned immediately
    img = mcrt.TileRenderer.render(mcrt.MeshBuilder.buildDefaultScene(), abi.Config(width=8, height=8, tileSize=0))
    assert mcrt.TileRenderer.lastErrors() == [] and (img[..., 3] == 1).all() and (img[..., :3] == 0).all()


# ---- flattener ------------------------------------------------------------------------------------
def test_flatten_blob_layout(mcrt):
    sd = scenes.skin_scene("S64", 6)
    blob = mcrt.flatten(sd)
    hdr = np.frombuffer(blob[:16], np.uint32)
    assert hdr[0] == 0x4D435254 and hdr[1] == 12 and hdr[2] == 12 * 272  # 12 meshes, 3264 texels
    assert len(blob) == 192 + 12 * 192 + 3264 * 16 + 3264 // 16 * 4
    f = np.frombuffer(blob[:192], np.float32)
    assert np.allclose(f[12:15], [0, 18, 50]) and abs(f[15] - np.tan(np.radians(30.0))) < 1e-6  # camera pos, tan(fov/2)
    assert np.allclose(f[16:19], [0, 0, -1]) and np.allclose(f[20:23], [1, 0, 0]) and np.allclose(f[24:27], [0, 1, 0])
    mesh0 = np.frombuffer(blob[192:192 + 192], np.float32)
    assert np.allclose(mesh0[0:6], [-4, 24, -4, 4, 32, 4])  # head AABB (local space: the mesh is posed)
    flags = np.frombuffer(blob[192 + 68:192 + 72], np.uint32)[0]
    assert flags & 2 and flags & 4 and flags & 8 and not flags & 1  # rotated, X and Z applied, inner


def test_flatten_first_pass_groups(mcrt):
    """Every mesh is in exactly one group; a member's box lies inside its root's box; posed meshes are
    their own roots (their boxes live in different frames)."""
    for pose, expect_roots in ((0, 6), (6, None)):
        blob = mcrt.flatten(scenes.skin_scene("S64", pose))
        hdr = np.frombuffer(blob[:192], np.uint32)
        n = int(hdr[1])
        roots = int(hdr[37]) | (int(hdr[38]) << 32)
        seen = 0
        for i in range(n):
            rec = blob[192 + 192 * i:192 + 192 * (i + 1)]
            f = np.frombuffer(rec, np.float32)
            u = np.frombuffer(rec, np.uint32)
            group = int(u[44]) | (int(u[45]) << 32)
            rotated = bool(u[17] & 2)
            if not (roots >> i) & 1:
                assert group == 0
                continue
            assert group & (1 << i) and not (seen & group)
            seen |= group
            if rotated:
                assert group == 1 << i
            for j in range(n):
                if (group >> j) & 1 and j != i:
                    g = np.frombuffer(blob[192 + 192 * j:192 + 192 * (j + 1)], np.float32)
                    assert (f[0:3] <= g[0:3]).all() and (f[3:6] >= g[3:6]).all()
        assert seen == (1 << n) - 1
        if expect_roots is not None:
            assert bin(roots).count("1") == expect_roots  # each outer-layer box encloses its inner box


def test_flatten_rejects_malformed_scenes(mcrt):
    from minecraftskin_raytracer_amd._lib import McrtError

    tex = abi.Texture(4, 4, np.ones((3, 4), np.float32))  # fewer pixels than w*h
    sc = scenes.simple_scene([scenes.build_box(tex, (0, 0, 0), (2, 2, 2))])
    with pytest.raises(McrtError):
        mcrt.flatten(mcrt.SceneDesc(sc))


def test_flatten_texture_edge_cases(mcrt):
    empty = abi.Texture(0, 0, np.zeros((0, 4), np.float32))
    box = scenes.build_box({"back": None, "front": empty, "left": scenes.solid((1, 1, 1, 1)), "right": None, "top": None, "bottom": None},
                           (0, 0, 0), (2, 2, 2))
    blob = mcrt.flatten(mcrt.SceneDesc(scenes.simple_scene([box])))
    tex_off = np.frombuffer(blob[192 + 72:192 + 96], np.int32)
    assert tex_off.tolist() == [-1, -2, 0, -1, -1, -1]  # nullptr → magenta, empty → Color(), pooled


# ---- scene builder vs the reference-generated fixture ----------------------------------------------
@pytest.mark.parametrize("name,kind,pose", [("S64_pose6", "S64", 6), ("S32_pose1", "S32", 1)])
def test_scene_builder_matches_golden(mcrt, name, kind, pose):
    g = np.load(os.path.join(GOLDEN, f"scene_{name}.npz"))
    d = scenes.skin_scene(kind, pose).to_numpy()
    assert len(d["meshes"]) == int(g["n_meshes"]) and len(d["textures"]) == int(g["n_textures"])
    for i, m in enumerate(d["meshes"]):
        for k, v in m.items():
            assert np.asarray(v).tobytes() == g[f"mesh{i}_{k}"].tobytes(), (i, k)
    for i, t in enumerate(d["textures"]):
        assert [t["width"], t["height"]] == g[f"tex{i}_wh"].tolist()
        assert t["pixels"].tobytes() == g[f"tex{i}_px"].tobytes()
    for k in ("light_position", "light_color", "camera_position", "camera_target", "camera_up", "background_color",
              "light_intensity", "light_radius", "camera_fov"):
        assert np.asarray(d[k]).tobytes() == g[k].tobytes(), k


def test_builtin_poses_and_mesh_counts(mcrt):
    poses = mcrt.getBuiltinPoses()
    assert len(poses) == 7 and poses[0].tolist() == [0.0] * 12
    assert poses[6].tolist() == [30, 15, 0, 5, -45, 30, 150, -10, 0, 0, 0, 0]
    assert len(scenes.skin_scene("S64", 0).to_numpy()["meshes"]) == 12  # inner + non-empty outer layers
    assert len(scenes.skin_scene("S32", 0).to_numpy()["meshes"]) == 7   # legacy: only the head has an outer layer
    assert len(mcrt.MeshBuilder.buildDefaultScene().to_numpy()["meshes"]) == 6
    with pytest.raises(ValueError):
        mcrt.MeshBuilder.buildScene(np.zeros((48, 64, 4), np.uint8))


def test_synthetic_skin_is_deterministic(mcrt):
    a = mcrt.synthetic_skin("S64")
    assert a.shape == (64, 64, 4) and a.dtype == np.uint8
    assert a[0, 0].tolist() == [5, 4, 139, 255]  # first three LCG steps from seed 12345, >> 24
    assert (a[16:32, :, 3] == 255).all() and 0.25 < (a[32:48, :, 3] == 255).mean() < 0.5
    assert mcrt.synthetic_skin("S32").shape == (32, 64, 4)


def test_bounce_limit_is_checked_before_any_device_work(mcrt):
    """max_bounces above 4000 is refused with MCRT_ERR_INVALID (one stack slot per level and sample; the
    reference itself stops at the first miss, raytracer.cpp:94-102) — by every host-buffer entry point,
    before a device is touched."""
    import ctypes as C

    import scenes
    from minecraftskin_raytracer_amd import _lib

    lib = _lib.load()
    sd = scenes.skin_scene("S64", 0)
    cfg = abi.Config(width=16, height=16, maxBounces=4001).to_c()
    out = np.zeros((16, 16, 4), np.float32)
    assert lib.mcrt_render(sd.ptr, C.byref(cfg), abi.fptr(out), C.cast(None, abi.PROGRESS_FN), None, 0) == abi.MCRT_ERR_INVALID
    assert b"4000" in lib.mcrt_last_error()
    assert lib.mcrt_render_multi(sd.ptr, C.byref(cfg), abi.fptr(out), C.cast(None, abi.PROGRESS_FN), None, None, 0, 0) == abi.MCRT_ERR_INVALID
    assert lib.mcrt_render_tile(sd.ptr, C.byref(cfg), 0, abi.fptr(out), 0) == abi.MCRT_ERR_INVALID
    assert lib.mcrt_render_png(sd.ptr, C.byref(cfg), b"/tmp/never_written.png", 0) == abi.MCRT_ERR_INVALID


def test_scene_desc_pointer_keeps_its_owner_alive(oracle):
    """`.ptr` of a temporary description stays valid for as long as the pointer is referenced (round 1 crashed in
    the oracle when a test passed `skin_scene(...).ptr` and the description had been collected)."""
    import gc

    p = scenes.skin_scene("S64", 3).ptr  # the SceneDesc itself is a temporary
    q = scenes.skin_scene("S32", 1).ptr
    gc.collect()
    assert p.contents.n_meshes == 12 and q.contents.n_meshes == 7
    img = oracle.render(p, abi.Config(width=24, height=18, maxBounces=1))
    assert img.shape == (18, 24, 4) and np.isfinite(img).all()


def test_render_entry_points_without_a_device(mcrt):
    """No HIP device: every way of naming devices records ONE TileError{-1, ...} and returns the Color() image
    (tile_renderer.cpp:158-166: render() never throws for render failures)."""
    if mcrt.device_count() > 0:
        pytest.skip("for boxes without a GPU")
    sd = scenes.skin_scene("S64", 0)
    cfg = abi.Config(width=32, height=16)
    for device in (0, "all", -1, [0, 0]):
        img = mcrt.TileRenderer.render(sd, cfg, device=device)
        errs = mcrt.TileRenderer.lastErrors()
        assert len(errs) == 1 and errs[0][0] == -1 and "device" in errs[0][1].lower(), (device, errs)
        assert img.shape == (16, 32, 4) and (img[..., :3] == 0).all() and (img[..., 3] == 1).all()


def test_render_out_argument_is_validated(mcrt):
    sd = scenes.skin_scene("S64", 0)
    cfg = abi.Config(width=32, height=16)
    for bad in (np.zeros((16, 32, 3), np.float32), np.zeros((16, 32, 4), np.float64), np.zeros((32, 16, 4), np.float32),
                np.zeros((16, 64, 4), np.float32)[:, ::2]):
        with pytest.raises(ValueError):
            mcrt.TileRenderer.render(sd, cfg, out=bad)
    with pytest.raises(ValueError):
        mcrt.TileRenderer.render(sd, cfg, device="some")


def test_device_argument_normalisation():
    """TileRenderer.render's `device`: any integer type is an index, -1 / "all" is every device, other negatives and
    unknown strings are refused, a sequence lists one rank per entry."""
    import numpy as np
    import pytest
    from minecraftskin_raytracer_amd.api import _devices

    assert _devices(0) == (0, None) and _devices(np.int64(2)) == (2, None) and _devices(np.uint8(1)) == (1, None)
    assert _devices(-1) == (None, []) and _devices("all") == (None, [])
    assert _devices([0, np.int32(1), 0]) == (None, [0, 1, 0]) and _devices(()) == (None, [])
    for bad in (-2, "gpu0", [0, -1], 1.5):
        with pytest.raises((ValueError, TypeError)):
            _devices(bad)
