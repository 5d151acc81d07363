"""Per-kernel HBM traffic and VALU instructions of the LAST frame in a tools/pmc_run.sh dispatch table.
    python tools/pmc_frame.py <dispatches.txt>
HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 counts KiB; FETCH doubled per the gfx950 correction of
MI355X_MICROARCH.md)."""
import json, sys
rows, hdr = [], None
for l in open(sys.argv[1]).read().splitlines():
    if l.startswith('idx kernel'):
        hdr = l.split()[2:]
    elif hdr and l and l[0].isdigit():
        parts = l.split()
        n = len(hdr)
        rows.append((' '.join(parts[1:-n]), dict(zip(hdr, map(float, parts[-n:])))))
# the last COMPLETE frame: from a plan_tiles to the resolve that follows it
ends = [i for i, (n, d) in enumerate(rows) if 'resolve' in n]
end = ends[-1]
start = max(i for i, (n, d) in enumerate(rows[:end]) if 'plan_tiles' in n)
first = start - 1 if start > 0 and 'fillBuffer' in rows[start - 1][0] else start  # (older builds cleared the counters ahead of plan_tiles)
frame = rows[first:end + 1]
tf = tw = tv = 0.0
for n, d in frame:
    f, w, v = d.get('FETCH_SIZE', 0), d.get('WRITE_SIZE', 0), d.get('SQ_INSTS_VALU', 0)
    tf, tw, tv = tf + f, tw + w, tv + v
    print(f"{n:30s} fetch x2 {2 * f * 1.024 / 1e3:7.1f} MB   write {w * 1.024 / 1e3:7.1f} MB   VALU wave-instructions {v:.3e}")
total = (2 * tf + tw) * 1024
print(f"frame: {total / 1e6:.1f} MB counted HBM traffic (fetch x2 {2 * tf * 1.024 / 1e3:.1f} + write {tw * 1.024 / 1e3:.1f}), VALU {tv:.4e}")
print(json.dumps({"hbm_bytes": int(total), "valu_wave_instructions": int(tv)}))
