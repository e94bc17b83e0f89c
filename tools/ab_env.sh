#!/bin/bash
# SAME-BOX A/B of one environment switch of the library (read once per process): tools/ab_env.sh VAR rounds script [args] -- runs the script
# alternately with VAR=0 and VAR=1, `rounds` times, and prints every run's lines that contain "ms" or "captions/s" behind the value
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
var=$1; rounds=$2; shift 2
for r in $(seq 1 $rounds); do
  for v in 0 1; do
    env $var=$v python3 "$@" 2>&1 | grep -E "ms|captions/s" | sed "s/^/$var=$v  /"
  done
done
