#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
export ROUND=r04b
bash tools/prof_any.sh aoa_beam5_b64 tools/perf_aoa_beam.py 64 2>&1 | head -14
timeout -k 10 900 python3 -m pytest tests/test_gpu_aoa.py tests/test_gpu_aoa_adaptive.py tests/test_gpu_butd.py -x -q 2>&1 | tail -4
timeout -k 10 300 python3 tools/perf_eval.py 128 2>&1 | grep -v amdgpu | tail -3
