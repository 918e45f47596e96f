#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT; : > $OUT/smallm.txt
run() { env "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f img/s  igemm %.0f TF/s  kernel ms %s' % (d['value'], d['roofline']['achieved'], d['kernel_time_ms_est']))"; }
for r in 1 2; do for v in 0 1; do
  echo "cn0 p8 SMALL_M=$v: $(run FGDM_IGEMM_SMALL_M=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 0 --prompts 8)" | tee -a $OUT/smallm.txt
done; done
for v in 0 1; do
  echo "cn2 p8 SMALL_M=$v: $(run FGDM_IGEMM_SMALL_M=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 2 --prompts 8)" | tee -a $OUT/smallm.txt
  echo "cn1 p16 SMALL_M=$v: $(run FGDM_IGEMM_SMALL_M=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage)" | tee -a $OUT/smallm.txt
done
