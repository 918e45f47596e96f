#!/bin/bash
# 64 x 160 tiles emitting the LayerNorm partial sums themselves (FGDM_IGEMM_STATS64=0: the separate row_stats pass as before): tests, then A/B
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_blocks.py -x -q -m gpu -k "layernorm or geglu or transformer or Transformer or block" > $OUT/stats64_tests.log 2>&1 || { tail -30 $OUT/stats64_tests.log; exit 1; }
tail -3 $OUT/stats64_tests.log
for r in 1 2; do
  for v in 0 1; do
    for cfg in "0 8" "2 8" "1 16"; do
      set -- $cfg
      FGDM_IGEMM_STATS64=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --prompts $2 --controlnets $1 --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('stats64=$v cn$1 p$2: %.3f img/s  igemm %.0f TF/s' % (d['value'], d['roofline']['achieved']))" || exit 1
    done
  done
done | tee $OUT/stats64_ab.txt
