#!/bin/bash
# same-box A/B of the bench headline: tools/ab_bench.sh "VAR=val VAR2=val" "..." ...   (each variant: 2 runs of 20 steps)
for v in "$@"; do
  for rep in 1 2; do
    r=$(env $v python bench.py --headline-only --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read()); print('%.0f captions/s %.3f ms' % (j['value'], j['ms_per_step']))")
    echo "[$v] $r"
  done
done
