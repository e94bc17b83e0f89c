#!/bin/bash
# usage: tools/prof_any.sh <tag> <python script and args...>  -> gpurun_out/prof_r03/<tag>_kernel_stats.csv + top kernels on stdout
tag=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PD=$(mktemp -d /tmp/prof_XXXXXX)
script=$1; shift; rocprofv3 --kernel-trace --stats --output-format csv -d $PD -- python3 $ROOT/$script "$@" > $OUT/${tag}.log 2> $OUT/${tag}.err
cp $(find $PD -name '*kernel_stats.csv' | head -1) $OUT/${tag}_kernel_stats.csv
tail -3 $OUT/${tag}.log
python3 - $OUT/${tag}_kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:22]:
    print("%-84s %6s %9.1f %6.2f%%" % (r["Name"][:84], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("total kernel ms", tot / 1e6)
PY
