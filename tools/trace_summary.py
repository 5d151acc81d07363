"""Print per-kernel stats and the last frame's timeline from a rocprofv3 --kernel-trace --stats run."""
import csv, glob, sys
root = sys.argv[1]
nlast = int(sys.argv[2]) if len(sys.argv) > 2 else 22
f = glob.glob(root + '/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'].split('(')[0][-40:]:40s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.2f} pct={float(r['Percentage']):6.2f} min={float(r['MinNs'])/1e3:8.2f} max={float(r['MaxNs'])/1e3:8.2f}")
t = glob.glob(root + '/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(t)), key=lambda r: int(r['Start_Timestamp']))
last = rows[-nlast:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print(f"{r['Kernel_Name'].split('(')[0][-34:]:34s} start={(int(r['Start_Timestamp'])-t0)/1e3:8.1f}us dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f}us grid={r['Grid_Size_X']:>8s} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']}")
