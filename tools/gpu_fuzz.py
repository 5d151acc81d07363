"""Long randomised parity sweep: HIP path vs the CPU oracle, bit-exact.  usage: gpu_fuzz.py <first_seed> <count> [bundle|wide]
(bundle: the cases aimed at the whole-bundle shadow decisions, fuzz_cases.make_bundle_case; wide: the same scenes scaled by
1e-5 ... 1e6 or moved up to 3e6 away from the origin, fuzz_cases.make_wide_case)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import minecraftskin_raytracer_amd as M
import oraclelib
from fuzz_cases import make_bundle_case, make_case, make_wide_case

first, count = int(sys.argv[1]), int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "bundle":
    make_case = make_bundle_case
if len(sys.argv) > 3 and sys.argv[3] == "wide":
    make_case = make_wide_case
orc = oraclelib.Oracle()
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    sd, cfg, what = make_case(seed)
    img = M.TileRenderer.render(sd, cfg)
    ref = orc.render(sd.ptr, cfg)
    errs = M.TileRenderer.lastErrors()
    same = np.array_equal(img.view(np.uint32), ref.view(np.uint32)) or (np.isnan(img) == np.isnan(ref)).all() and np.array_equal(np.nan_to_num(img), np.nan_to_num(ref))
    if errs or not same:
        bad += 1
        diff = int((img.view(np.uint32) != ref.view(np.uint32)).sum())
        print(f"MISMATCH {what}: errors={errs} differing floats={diff} max abs diff={np.nanmax(np.abs(img - ref))}", flush=True)
    if (seed - first) % 50 == 49:
        print(f"... {seed - first + 1} cases, {bad} mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"fuzz: {count} cases from seed {first}: {bad} mismatch(es)")
sys.exit(1 if bad else 0)
