#!/bin/bash
# round 4: the three GEMMs of a beam step (640 rows; TD gates, LM gates, vocabulary projection with FIXED weights, cycled as a decode
# does) on the shipped 128 x 128 split-precision kernel and on the planes GEMM with weight planes packed once: per-kernel times
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
run() {   # tag, env value, splits
  d=$(mktemp -d /tmp/prof_XXXXXX)
  ICZ_GEMM_PLANES_STATIC_W=1 ICZ_GEMM_PLANES_TEST=$2 rocprofv3 --kernel-trace --output-format csv -d $d -- python3 $ROOT/tools/perf_beam_gemm.py $3 > $d/out.txt 2>&1
  f=$(find $d -name '*kernel_trace.csv' | head -1)
  echo "== $1 cfg $2 splits $3 : $(grep 'rel err' $d/out.txt | tr '\n' ' ')"
  python3 - "$f" <<'PY'
import csv, sys, statistics, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "gemm" in n or "slab" in n or "pack" in n:
        d[(n.split("(")[0][-44:], r["Grid_Size_X"] + "x" + r.get("Grid_Size_Y", "1") + "x" + r.get("Grid_Size_Z", "1"))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(d.items()):
    if len(v) >= 30:
        print("   %-46s grid %-16s n %4d median %8.2f us" % (k[0], k[1], len(v), statistics.median(v)))
        tot += statistics.median(v)
print("   sum of medians per step: %.1f us" % tot)
PY
}
run shipped 0 "0 0 0"
run planes 3 "3 3 1"
run planes 2 "6 6 1"
run planes 2 "6 6 2"
run planes 3 "6 3 1"
run planes 1 "4 5 2"
