"""dev tool: per-launch durations of tools/perf_gemm_fixed.py from a rocprofv3 kernel trace CSV -> fixed + per-stage cost."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
g = [r for r in rows if "gemm_nt" in r["Kernel_Name"]]
per = collections.OrderedDict()
ks = (512, 1024, 2048, 4096, 8192)
for i, r in enumerate(g):
    k = ks[min(i // 30, 4)]
    per.setdefault(k, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
prev = None
for k, v in per.items():
    v = sorted(v[5:])
    med = v[len(v) // 2]
    st = k // 128 // NS
    msg = "K %5d (%2d stages per workgroup): median %.2f us" % (k, st, med)
    if prev:
        msg += "   -> %.2f us per stage" % ((med - prev[1]) / (st - prev[0]))
    print(msg)
    prev = (st, med)
