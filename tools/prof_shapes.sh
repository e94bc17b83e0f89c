#!/bin/bash
# usage: tools/prof_shapes.sh <tag> "M N K ns" ...   -> prints avg kernel duration of the GEMM kernels per shape
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for shape in "$@"; do
  d=$(mktemp -d /tmp/prof_XXXXXX)
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $ROOT/tools/perf_skinny_one.py $shape > /dev/null 2>&1
  f=$(find $d -name '*kernel_stats.csv' | head -1)
  echo "== $tag shape $shape"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "gemm" in n or "slab" in n or "pack" in n:
        print("   %-70s calls %4s avg %8.2f us min %8.2f" % (n[:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
done
