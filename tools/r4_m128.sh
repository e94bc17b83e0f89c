#!/bin/bash
# round 4: the 128-row resident GEMM stand-alone -- float64 bound, then per-launch kernel time against two 64-row launches and
# against the 128 x 128-tile kernel (ICZ_GEMM_RESIDENT_M128=0) at the three decoder-step shapes
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
mkdir -p gpurun_out
timeout -k 10 600 python3 -m pytest tests/test_gpu_butd.py -k gemm_against_float64 -x -q 2>&1 | tail -5 || exit 1
for sh in "64 4096 4096 0" "128 4096 4096 0" "64 4096 3072 0" "128 4096 3072 0" "64 10112 1024 0" "128 10112 1024 0" "100 4096 4096 0"; do
  tools/prof_shapes.sh m128 "$sh"
done
export ICZ_GEMM_RESIDENT_M128=0
for sh in "128 4096 4096 0" "128 4096 3072 0" "128 10112 1024 0"; do
  tools/prof_shapes.sh tile128 "$sh"
done
