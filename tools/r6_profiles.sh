#!/bin/bash
# round-6 evidence in one gpurun call: kernel stats of the bench command (+ the two PMC passes), beam 5 x 128, BUTDSpatial XE, the AoA
# SCST step, AoA beam 5 x 64 -> gpurun_out/prof_r06/ (what is judged is copied into profiles/r06_*), then the bench line itself
export ROUND=r06
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out/prof_r06
bash tools/collect_profiles.sh > gpurun_out/prof_r06/collect.log 2>&1
bash tools/prof_any.sh beam5_b128 tools/perf_eval.py 128 > gpurun_out/prof_r06/beam.top 2>&1
bash tools/prof_any.sh xe_spatial49 tools/perf_xe_spatial.py > gpurun_out/prof_r06/xe.top 2>&1
bash tools/prof_aoa_engine.sh > gpurun_out/prof_r06/aoa.top 2>&1
bash tools/prof_any.sh aoa_beam5_b64 tools/perf_aoa_beam.py 64 > gpurun_out/prof_r06/aoa_beam.top 2>&1
timeout -k 10 900 python3 bench.py > gpurun_out/prof_r06/bench_line.json 2> gpurun_out/prof_r06/bench.err
ls gpurun_out/prof_r06
tail -c 600 gpurun_out/prof_r06/bench_line.json
# round 5 additions: the SCST step of a model that ends its captions (early-out of the steps behind the reference's break) and of
# an 8-image batch (merged greedy + sampled chain)
ICZ_PERF_BREAK=11 ICZ_PERF_ROUNDS=1 bash tools/prof_any.sh scst_end_biased tools/perf_options.py early_out=1 > gpurun_out/prof_r06/end_biased.top 2>&1
ICZ_PERF_B=8 bash tools/prof_any.sh scst_b8 tools/perf_small_one.py > gpurun_out/prof_r06/b8.top 2>&1
