"""Summarise rocprofv3 --pmc CSVs: per kernel, per counter, mean value per dispatch."""
import csv, glob, os, sys
from collections import defaultdict
root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(root, "pass*", "**", "*counter_collection.csv"), recursive=True):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "?").split("(")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(acc):
    print(f"== {k}")
    for c in sorted(acc[k]):
        v = acc[k][c]
        print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
