#!/bin/bash
# round-3 baseline: stamps of the resident kernel at the three decoder-step shapes, greedy chain under rocprofv3, phases
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/r3
mkdir -p $OUT
cd $ROOT
for shape in "64 4096 3072 12" "64 4096 4096 16" "64 10112 1024 4"; do
  echo "== stamps $shape" >> $OUT/stamps.log
  RESIDENT=1 ICZ_SKINNY_ABL=4 timeout -k 10 120 python3 tools/perf_skinny_stamps.py $shape >> $OUT/stamps.log 2>&1
done
timeout -k 10 200 python3 tools/perf_greedy.py 64 > $OUT/greedy.log 2>&1
timeout -k 10 300 python3 tools/perf_phases.py > $OUT/phases.log 2>&1
timeout -k 10 300 bash tools/prof_any.sh r3_greedy tools/perf_greedy.py 64 > $OUT/prof_greedy.log 2>&1
cat $OUT/stamps.log $OUT/greedy.log $OUT/phases.log
