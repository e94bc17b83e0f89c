#!/bin/bash
# rocprofv3 summaries of the bench command -> gpurun_out/prof_r03/ (copy what is to be judged into profiles/)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_r03
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_stats /tmp/p_fetch /tmp/p_write
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-h2d > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
cp $(find /tmp/p_stats -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/p_fetch -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/fetch.err
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/p_write -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-h2d > /dev/null 2> $OUT/write.err
echo "write done"
python3 $ROOT/tools/pmc_summary.py $(find /tmp/p_fetch -name '*counter_collection.csv' | head -1) $(find /tmp/p_write -name '*counter_collection.csv' | head -1) $OUT/pmc_traffic.json
ls -la $OUT
