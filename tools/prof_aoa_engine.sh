#!/bin/bash
# rocprofv3 kernel stats of tools/perf_aoa_engine.py (13 AoA SCST steps through the Engine) -> gpurun_out/prof_r03/aoa_kernel_stats.csv
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PD=$(mktemp -d /tmp/prof_XXXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $PD -- python3 $ROOT/tools/perf_aoa_engine.py > $OUT/aoa_engine.log 2> $OUT/aoa_stats.err
cp $(find $PD -name '*kernel_stats.csv' | head -1) $OUT/aoa_kernel_stats.csv
tail -2 $OUT/aoa_engine.log
