#!/bin/bash
# twins' single-pass GroupNorm launches grouped (FGDM_GN_GROUP): tests, then A/B on the default bench and the 8-prompt configurations
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_ops.py -x -q -m gpu -k "grouped or groupnorm or several or full_width" > $OUT/gngroup_tests.log 2>&1 || { tail -30 $OUT/gngroup_tests.log; exit 1; }
tail -3 $OUT/gngroup_tests.log
for r in 1 2; do
  for v in 0 1; do
    for cfg in "1 16" "2 8"; do
      set -- $cfg
      FGDM_GN_GROUP=$v timeout -k 10 300 python bench.py --steps 2 --warmup 1 --prompts $2 --controlnets $1 --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('gn_group=$v cn$1 p$2: %.3f img/s  norm est %.1f ms' % (d['value'], d['kernel_time_ms_est']['norm']))" || exit 1
    done
  done
done | tee $OUT/gngroup_ab.txt
