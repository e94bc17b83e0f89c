#!/bin/bash
# rocprofv3 summaries of the bench command -> gpurun_out/prof_r03/ (copy what is to be judged into profiles/)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_${ROUND:-r04}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp ICZ_BENCH_ROOFLINE_TOL=10      # (under the profiler the live event pairs are not the judged number)
PD=$(mktemp -d /tmp/prof_XXXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $PD/stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find $PD/stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $PD/fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $PD/write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/write.err
echo "write done"
python3 $ROOT/tools/pmc_summary.py $(find $PD/fetch -name '*counter_collection.csv' | head -1) $(find $PD/write -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic.json
ls -la $OUT
