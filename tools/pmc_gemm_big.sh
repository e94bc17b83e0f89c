#!/bin/bash
# round 5: SQ counters of the many-row split-precision GEMM kernels on the weight-gradient shapes (tools/perf_gemm_big.py tn), one pass per
# kernel choice (ICZ_GEMM_BIG=0: 128 x 128 two-barrier kernel, 1: 256 x 256 eight waves, 4: 128 x 128 three per CU) -> gpurun_out/pmc_big/
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_big
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in 0 1 4; do
  PD=$(mktemp -d /tmp/pmcbig_XXXXXX)
  ICZ_GEMM_BIG=$c rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE \
      --kernel-trace --output-format csv -d $PD -- python3 $ROOT/tools/perf_gemm_big.py tn > $OUT/run_$c.log 2> $OUT/run_$c.err
  cp $(find $PD -name '*counter_collection.csv' | head -1) $OUT/counters_$c.csv 2>/dev/null
  cp $(find $PD -name '*kernel_trace.csv' | head -1) $OUT/trace_$c.csv 2>/dev/null
done
python3 - <<PY
import csv, collections, glob
for c in (0, 1, 4):
    try:
        rows = list(csv.DictReader(open("$OUT/counters_%d.csv" % c)))
    except Exception as e:
        print(c, "no counters", e); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in rows:
        k = r["Kernel_Name"]
        if "gemm_big_x3" not in k and "tn128_x3" not in k: continue
        key = (k.split("(")[0][-70:], r.get("Grid_Size"), )
        agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[key] += 1
    for key, v in agg.items():
        d = n[key] or 1
        print("cfg %d %-72s grid %-8s launches %3d  MFMA busy/SQ busy %.3f  LDS conflict/LDS active %.3f  wait_any %.2f wait_inst %.2f active_inst %.2f of wave cycles (x4)" % (
            c, key[0], key[1], d, v["SQ_VALU_MFMA_BUSY_CYCLES"] / max(v["SQ_BUSY_CYCLES"], 1), v["SQ_LDS_BANK_CONFLICT"] / max(v["SQ_LDS_IDX_ACTIVE"], 1),
            v["SQ_WAIT_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_WAIT_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1), v["SQ_ACTIVE_INST_ANY"] / max(v["SQ_WAVE_CYCLES"], 1)))
PY
