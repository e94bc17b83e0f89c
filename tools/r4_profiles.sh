#!/bin/bash
# round-4 evidence in one gpurun call: kernel stats of the bench command (+ the two PMC passes), beam 5 x 128, BUTDSpatial XE, the AoA
# SCST step, AoA beam 5 x 64 -> gpurun_out/prof_r04/ (what is judged is copied into profiles/r04_*), then the bench line itself
export ROUND=r04
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/prof_r04
bash tools/collect_profiles.sh > gpurun_out/prof_r04/collect.log 2>&1
bash tools/prof_any.sh beam5_b128 tools/perf_eval.py 128 > gpurun_out/prof_r04/beam.top 2>&1
bash tools/prof_any.sh xe_spatial49 tools/perf_xe_spatial.py > gpurun_out/prof_r04/xe.top 2>&1
bash tools/prof_aoa_engine.sh > gpurun_out/prof_r04/aoa.top 2>&1
bash tools/prof_any.sh aoa_beam5_b64 tools/perf_aoa_beam.py 64 > gpurun_out/prof_r04/aoa_beam.top 2>&1
timeout -k 10 900 python3 bench.py > gpurun_out/prof_r04/bench_line.json 2> gpurun_out/prof_r04/bench.err
ls gpurun_out/prof_r04
tail -c 600 gpurun_out/prof_r04/bench_line.json
