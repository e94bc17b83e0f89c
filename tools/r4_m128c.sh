#!/bin/bash
# round 4: tuned 128-row resident GEMM: float64 bound, kernel time per shape, stamps, and what a 128-row decode chain costs against
# the concurrent pair of 64-row chains (tools/perf_merge.py)
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_butd.py -k gemm_against_float64 -x -q 2>&1 | tail -3 || exit 1
for sh in "128 4096 4096 0" "128 4096 3072 0" "128 10112 1024 0"; do
  tools/prof_shapes.sh m128 "$sh"
done
cp simpleimagecaptionzoo_amd/libicz.so /tmp/libicz_keep.so
cp tools/ab/libicz_dev.so simpleimagecaptionzoo_amd/libicz.so
for sh in "128 4096 4096" "128 10112 1024"; do
  timeout -k 10 120 python3 tools/perf_m128_stamps.py $sh 2>&1 | grep -v amdgpu
done
cp /tmp/libicz_keep.so simpleimagecaptionzoo_amd/libicz.so
echo "== perf_merge, m128 kernel"
timeout -k 10 300 python3 tools/perf_merge.py 2>&1 | grep -v amdgpu
echo "== perf_merge, ICZ_GEMM_RESIDENT_M128=0 (128 x 128 tile kernel at 128 rows)"
ICZ_GEMM_RESIDENT_M128=0 timeout -k 10 300 python3 tools/perf_merge.py 2>&1 | grep -v amdgpu
