#!/bin/bash
# round 4: new tests (full-width goldens, rank rehearsals, bench self-spawn) + the activation-through-L2 microbenchmark
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
echo "== microbenchmark"; timeout -k 10 120 tools/cxx/act_through_l2 2>&1 | grep -v amdgpu
echo "== round4 tests"; timeout -k 10 600 python3 -m pytest tests/test_gpu_round4.py -x -q 2>&1 | tail -15
echo "== dist tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_dist_two_ranks.py -x -q 2>&1 | tail -15
echo "== bench spawn tests"; timeout -k 10 1100 python3 -m pytest tests/test_gpu_round3.py -x -q -k "bench" 2>&1 | tail -15
