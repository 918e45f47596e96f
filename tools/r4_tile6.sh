#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "linear or layernorm" > $OUT/tile6_tests.log 2>&1 || { tail -30 $OUT/tile6_tests.log; exit 1; }
tail -3 $OUT/tile6_tests.log
timeout -k 10 300 python tools/bench_igemm.py --batch 16 --cfgs 0,10 --shapes "L2 lin 5120" 2>/dev/null
for r in 1 2; do
  for v in 0 1; do
    for cfg in "0 8" "2 8"; do
      set -- $cfg
      FGDM_IGEMM_SMALL_M=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --prompts $2 --controlnets $1 --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('small_m=$v cn$1 p$2: %.3f img/s  igemm %.0f TF/s' % (d['value'], d['roofline']['achieved']))" || exit 1
    done
  done
done | tee $OUT/tile6_ab.txt
