#!/bin/bash
# Round 4: per-GPU rates of the 8-prompt configurations (BASELINE configs[1], [3], [4] shard shapes); N-way grouping vs pairwise
OUT=gpurun_out/r4; mkdir -p $OUT; : > $OUT/cfgs.txt
run() { env "$@" 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%.3f img/s  igemm %.0f TF/s  whole path %.3f  kernel ms %s' % (d['value'], d['roofline']['achieved'], d['whole_path_mfma_frac'], d['kernel_time_ms_est']))"; }
echo "cn0 p8:  $(run FGDM_X=0 timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 0 --prompts 8)" | tee -a $OUT/cfgs.txt
for g in 2 5; do
  echo "cn2 p8 GROUP_MAX=$g:  $(run FGDM_GROUP_MAX=$g timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 2 --prompts 8)" | tee -a $OUT/cfgs.txt
  echo "cn3 p8 GROUP_MAX=$g:  $(run FGDM_GROUP_MAX=$g timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 3 --prompts 8)" | tee -a $OUT/cfgs.txt
done
