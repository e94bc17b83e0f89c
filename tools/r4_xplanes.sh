#!/bin/bash
# round 4: the 64-row resident GEMM fed with producer-written plane images (LDS-DMA) against the fp32 load-and-split prologue
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
echo "== float64 bound with plane images"
ICZ_GEMM_XPLANES_TEST=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_butd.py -k "gemm_against_float64 or resident_gemm" -x -q 2>&1 | tail -3 || exit 1
for sh in "64 4096 4096 0" "64 4096 3072 0" "64 10112 1024 0"; do
  tools/prof_shapes.sh fp32 "$sh"
  ICZ_GEMM_XPLANES_TEST=1 tools/prof_shapes.sh xplanes "$sh"
done
cp simpleimagecaptionzoo_amd/libicz.so /tmp/libicz_keep.so
cp tools/ab/libicz_dev.so simpleimagecaptionzoo_amd/libicz.so
for sh in "64 4096 4096" "64 10112 1024"; do
  timeout -k 10 120 python3 tools/perf_skinny_stamps.py $sh 2>&1 | grep -v amdgpu
  ICZ_GEMM_XPLANES_TEST=1 timeout -k 10 120 python3 tools/perf_skinny_stamps.py $sh 2>&1 | grep -v amdgpu
done
cp /tmp/libicz_keep.so simpleimagecaptionzoo_amd/libicz.so
