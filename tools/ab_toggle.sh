#!/bin/bash
# Same-box interleaved A/B of one environment toggle of the product (3 rounds x on/off, 30 timed steps each).
#   bash tools/ab_toggle.sh TSCD_PREFETCH_BACKBONE      (also: TSCD_FUSED_SGD, TSCD_LAP_PARALLEL, TSCD_LEVEL_STREAMS, TSCD_EAGER_CLASS_BRANCH)
var=${1:?name of a 0/1 environment toggle}
for r in 1 2 3; do
  for v in 1 0; do
    ms=$(env $var=$v timeout -k 10 200 python bench.py --steps 30 --warmup 8 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print('%.2f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "round $r $var=$v ms_per_step=$ms"
  done
done
