#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
for ng in 4 2; do
  echo "== FGDM_GN_REG_NG=$ng"
  FGDM_GN_REG_NG=$ng timeout -k 10 200 python tools/bench_norm.py 2>/dev/null | grep "groupnorm" | grep -v HW4096
done | tee $OUT/gnng.txt
for r in 1 2; do for ng in 4 2; do for cfg in "1 16" "0 8"; do
  set -- $cfg
  FGDM_GN_REG_NG=$ng timeout -k 10 300 python bench.py --steps 2 --warmup 1 --prompts $2 --controlnets $1 --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('gn NG<=$ng cn$1 p$2: %.3f img/s  norm est %.1f ms' % (d['value'], d['kernel_time_ms_est']['norm']))" || exit 1
done; done; done | tee -a $OUT/gnng.txt
