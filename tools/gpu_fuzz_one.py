"""Re-run single fuzz cases and show where the frames differ.  usage: gpu_fuzz_one.py [bundle] seed..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import minecraftskin_raytracer_amd as M
import oraclelib
from fuzz_cases import make_bundle_case, make_case

args = sys.argv[1:]
gen = make_case
if args and args[0] == "bundle":
    gen, args = make_bundle_case, args[1:]
orc = oraclelib.Oracle()
for seed in map(int, args):
    sd, cfg, what = gen(seed)
    img = M.TileRenderer.render(sd, cfg)
    ref = orc.render(sd.ptr, cfg)
    neq = (img.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(img) & np.isnan(ref))
    ys, xs = np.nonzero(neq.any(axis=2))
    print(what)
    print(f"  differing pixels: {len(ys)}", [(int(x), int(y), img[y, x].tolist(), ref[y, x].tolist()) for y, x in list(zip(ys, xs))[:4]], flush=True)
