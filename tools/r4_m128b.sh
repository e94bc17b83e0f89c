#!/bin/bash
# round 4: 128-row resident GEMM, gate shapes + stamps (DEV library swapped in on the box's scratch copy, then restored)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
for sh in "128 4096 4096 0" "128 4096 3072 0" "128 10112 1024 0" "100 4096 4096 0"; do
  tools/prof_shapes.sh m128 "$sh"
done
cp simpleimagecaptionzoo_amd/libicz.so /tmp/libicz_keep.so
cp tools/ab/libicz_dev.so simpleimagecaptionzoo_amd/libicz.so
for sh in "128 4096 4096" "128 4096 3072" "128 10112 1024"; do
  timeout -k 10 120 python3 tools/perf_m128_stamps.py $sh 2>&1 | grep -v amdgpu
done
for sh in "64 4096 4096" "64 10112 1024"; do
  timeout -k 10 120 python3 tools/perf_skinny_stamps.py $sh 2>&1 | grep -v amdgpu
done
cp /tmp/libicz_keep.so simpleimagecaptionzoo_amd/libicz.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_butd.py -k gemm_against_float64 -x -q 2>&1 | tail -3
