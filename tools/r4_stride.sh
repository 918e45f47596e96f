#!/bin/bash
# what do the live HIP-event brackets cost?  the default bench line at several sampling strides, alternating on one box
OUT=gpurun_out/r4; mkdir -p $OUT
for r in 1 2; do
  for st in 7 31 101; do
    timeout -k 10 300 python bench.py --steps 2 --warmup 1 --profile-stride $st --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('stride $st: %.3f img/s  %.1f ms/step  igemm %.1f TF/s over %d timed launches, avg %.2f us' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['timed_launches'], d['roofline']['avg_launch_us']))" || exit 1
  done
done | tee $OUT/stride_ab.txt
