#!/bin/bash
# kernel timeline of the LAST backward pass of a few eager SCST steps (ICZ_NO_GRAPHS=1: under the profiler a replayed graph runs its branches
# one after the other) -> gpurun_out/<tag>_bwd_timeline.txt   (tools/trace_timeline.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
tag=${1:-r06}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp ICZ_NO_GRAPHS=1
PD=$(mktemp -d /tmp/prof_XXXXXX)
rocprofv3 --kernel-trace --output-format csv -d $PD -- python3 $ROOT/tools/perf_headline.py 6 > $ROOT/gpurun_out/${tag}_bwd_trace.log 2>&1
f=$(find $PD -name "*kernel_trace.csv" | head -1)
python3 $ROOT/tools/trace_timeline.py $f > $ROOT/gpurun_out/${tag}_bwd_timeline.txt 2>&1
python3 $ROOT/tools/trace_timeline.py $f rollouts > $ROOT/gpurun_out/${tag}_rollouts_timeline.txt 2>&1
tail -3 $ROOT/gpurun_out/${tag}_bwd_trace.log
