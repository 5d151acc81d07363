"""Exhaustive check of rt::div_frame (rt_core.h): for every integer divisor d in [first, last] and EVERY float x a sample
coordinate can take (0 and 2^-33 .. d + 1), the frame-constant division equals the general (IEEE) division on the device.
usage: gpu_verify_div.py [first last]      (default 1 16384 = rt::kDivFrameMax; ~6e12 quotients, about a minute)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from minecraftskin_raytracer_amd import api

first, last = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1, 16384)
t0 = time.time()
total_bad = {False: 0, True: 0}
for two in (False, True):  # rt::div_frame (one correction: what the kernels use), then rt::div_frame2 (two corrections)
    for d0 in range(first, last + 1, 1024):
        n = min(1024, last + 1 - d0)
        bad, which = api.probe_div_const(d0, n, two)
        total_bad[two] += bad
        print(f"{'two corrections' if two else 'div_frame'}: divisors {d0} .. {d0 + n - 1}: {bad} mismatch(es)" + (f" (e.g. d = {which})" if bad else "") + f"   [{time.time() - t0:.0f} s]", flush=True)
print(f"rt::div_frame (one correction), divisors {first} .. {last}, every float in {{0}} and [2^-33, d + 1]: {total_bad[False]} mismatch(es)")
print(f"rt::div_frame2 (two corrections): {total_bad[True]} mismatch(es)")
sys.exit(1 if total_bad[False] else 0)
