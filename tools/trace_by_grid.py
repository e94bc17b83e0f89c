"""Reduce a rocprofv3 --kernel-trace CSV to median duration per (kernel, grid size): separates the shapes of one kernel."""
import csv, sys, glob, statistics, collections
files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
d = collections.defaultdict(list)
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0][-60:]
        d[(name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for k, v in sorted(d.items(), key=lambda kv: kv[0]):
    if pat in k[0]:
        print("%-62s grid %9s wg %4s  n %4d  median %9.1f us  min %9.1f" % (k[0], k[1], k[2], len(v), statistics.median(v), min(v)))
