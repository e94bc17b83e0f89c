#!/bin/bash
# kernel timeline (rocprofv3 --kernel-trace, csv with start / end stamps) of a few SCST steps with the weight gradients in N time chunks:
# tools/trace_bwd.sh <N> -> gpurun_out/r5/trace_c<N>/  (read with tools/trace_timeline.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export TMPDIR=/tmp ICZ_PERF_STEPS=6 ICZ_PERF_ROUNDS=1
n=$1
out=gpurun_out/r5/trace_c$n${TAG}
rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/perf_bwd_chunks.py $n > $out/run.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/trace_timeline.py $f > $out/timeline.txt 2>&1
find $out -name "*.csv" -size +20M -delete
tail -3 $out/run.log
