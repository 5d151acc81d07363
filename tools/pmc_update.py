"""Folds the counters of one workload's PMC run into profiles/pmc_traffic.json (what bench.py's `roofline` reads).
    python tools/pmc_update.py <workload> <dispatches.txt> <profile tag>
Per frame (the last complete one of the run): HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (rocprofv3 counts KiB; FETCH doubled
per the gfx950 correction of MI355X_MICROARCH.md), VALU wave-instructions = SQ_INSTS_VALU, both summed over the frame's dispatches,
plus the per-kernel split and the hash of the kernel sources the run was taken on (bench.py flags counters of other sources as stale).
A 4th argument stores the counters as a named sub-entry instead ("pipelined": the run was taken with MCRT_SHARED_GRIDS=1, the launch
shapes of a frame that shares the device — what bench.py's `value` runs)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

workload, path, tag = sys.argv[1], sys.argv[2], sys.argv[3]
variant = sys.argv[4] if len(sys.argv) > 4 else None  # "pipelined": counters of the launch shapes a frame takes when frames share the device
rows, hdr = [], None
for l in open(path).read().splitlines():
    if l.startswith('idx kernel'):
        hdr = l.split()[2:]
    elif hdr and l and l[0].isdigit():
        parts = l.split()
        n = len(hdr)
        rows.append((' '.join(parts[1:-n]), dict(zip(hdr, map(float, parts[-n:])))))
ends = [i for i, (n, d) in enumerate(rows) if 'resolve' in n]
end = ends[-1]
start = max(i for i, (n, d) in enumerate(rows[:end]) if 'plan_tiles' in n)
first = start - 1 if start > 0 and 'fillBuffer' in rows[start - 1][0] else start  # (older builds cleared the counters ahead of plan_tiles)
frame = rows[first:end + 1]
per_kernel, tf, tw, tv = {}, 0.0, 0.0, 0.0
for n, d in frame:
    f, w, v = d.get('FETCH_SIZE', 0.0), d.get('WRITE_SIZE', 0.0), d.get('SQ_INSTS_VALU', 0.0)
    tf, tw, tv = tf + f, tw + w, tv + v
    k = per_kernel.setdefault(n.split('<')[0].replace('_kernel', ''), {"hbm_bytes": 0, "valu_wave_instructions": 0})
    k["hbm_bytes"] += int((2 * f + w) * 1024)
    k["valu_wave_instructions"] += int(v)
out = os.path.join(ROOT, "profiles", "pmc_traffic.json")
data = json.load(open(out)) if os.path.exists(out) else {}
old = data.get(workload) if isinstance(data.get(workload), dict) else {}
entry = {"hbm_bytes": int((2 * tf + tw) * 1024), "valu_wave_instructions": int(tv), "profile": tag, "source_hash": bench.kernel_source_hash(), "per_kernel": per_kernel}
history = old.get("history", {})
if old.get("profile") and old.get("profile") != tag:
    history[old["profile"]] = {k: old[k] for k in ("hbm_bytes", "valu_wave_instructions") if k in old}
for k in ("before_bundle_decisions", "without_seed_table"):  # round 2's named predecessors
    if k in old:
        history[old[k].get("profile", k)] = {x: old[k][x] for x in ("hbm_bytes", "valu_wave_instructions")}
if history:
    entry["history"] = history
if variant:  # a sub-entry of the workload's entry (which must exist: run the plain pass first)
    sub = {k: entry[k] for k in ("hbm_bytes", "valu_wave_instructions", "source_hash", "per_kernel")}
    old[variant] = sub
    entry = old
data[workload] = entry
json.dump(data, open(out, "w"), indent=1)
print(json.dumps({workload: {k: entry[k] for k in ("hbm_bytes", "valu_wave_instructions", "source_hash")}}))
