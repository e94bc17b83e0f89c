#!/bin/bash
# round-3 evidence: kernel stats of the bench command (+ the two PMC passes), the beam / AoA / BUTDSpatial-XE tools -> gpurun_out/prof_r03/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
bash tools/collect_profiles.sh > gpurun_out_collect.log 2>&1; mkdir -p gpurun_out/prof_r03; mv gpurun_out_collect.log gpurun_out/prof_r03/collect.log
bash tools/prof_any.sh beam5_b128 tools/perf_eval.py 128 > gpurun_out/prof_r03/beam.top 2>&1
bash tools/prof_any.sh xe_spatial49 tools/perf_xe_spatial.py > gpurun_out/prof_r03/xe.top 2>&1
bash tools/prof_aoa_engine.sh > gpurun_out/prof_r03/aoa.top 2>&1
ls gpurun_out/prof_r03
