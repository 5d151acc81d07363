"""Per-kernel register / LDS / scratch use of the product library (hipcc -Rpass-analysis=kernel-resource-usage).

    python tools/kernel_resources.py            # rebuilds libmcrt.so verbosely and prints one line per kernel
"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, "-m", "minecraftskin_raytracer_amd.build", "--force", "--verbose"], cwd=ROOT,
                     capture_output=True, text=True)
if out.returncode != 0:
    sys.stderr.write(out.stderr)
    raise SystemExit(out.returncode)
cur = None
rows = {}
for line in out.stderr.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"\d+:\d+: remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for name, r in rows.items():
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(.*", "", dem).replace("void mcrt::", "").replace("mcrt::", "")
    print(f"{dem:42s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('SGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s}")
