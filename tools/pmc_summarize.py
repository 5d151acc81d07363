"""Summarise rocprofv3 --pmc CSVs.

    pmc_summarize.py <dir>                 per kernel, per counter: mean value per dispatch
    pmc_summarize.py <dir> --dispatches N  the first N dispatches in launch order, one row each,
                                           with the counters of every pass side by side
"""
import csv, glob, os, sys
from collections import defaultdict

root = sys.argv[1]
per_dispatch = int(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[2] == "--dispatches" else 0
acc = defaultdict(lambda: defaultdict(list))
seen = defaultdict(lambda: defaultdict(list))  # dispatch order index -> counter -> one value per PASS that collected it
names = {}
for f in sorted(glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True)):
    order = {}
    in_file = defaultdict(dict)  # a dispatch has several rows per counter (one per XCD / SE): summed within the pass
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
            d = int(row["Dispatch_Id"])
            if d not in order:
                order[d] = len(order)
            i = order[d]
            in_file[i][row["Counter_Name"]] = in_file[i].get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            names[i] = k
    for i, cs in in_file.items():
        for c, v in cs.items():
            seen[i][c].append(v)
# a counter collected in two passes is averaged, not added
rows = {i: {c: sum(v) / len(v) for c, v in cs.items()} for i, cs in seen.items()}
if per_dispatch:
    cols = sorted({c for r in rows.values() for c in r})
    print("idx kernel " + " ".join(cols))
    for i in sorted(rows)[:per_dispatch]:
        print(i, names[i].replace("void mcrt::", "").replace("mcrt::", "")[:28], " ".join(f"{rows[i].get(c, float('nan')):.4g}" for c in cols))
else:
    for k in sorted(acc):
        print(f"== {k}")
        for c in sorted(acc[k]):
            v = acc[k][c]
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
