#!/bin/bash
# round 4: the rollout pair free-running (ICZ_ROLLOUT_PACE=0) against paced variants (10 * record point + wait point), same box
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for round in 1 2; do
for m in 0 10 11 12 20 21 30 32 31 22 40 41; do
  echo "== pace $m (round $round)"
  ICZ_ROLLOUT_PACE=$m timeout -k 10 200 python3 tools/perf_phases.py 2>&1 | grep -E "rollouts|span"
done
done
