#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
for r in 1 2; do for tw in "" "--twin-streams"; do
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-first-stage $tw 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('twin_streams=%s: %.3f img/s  %.1f ms/step' % (d['twin_streams'], d['value'], d['ms_per_step']))" || exit 1
done; done | tee $OUT/twin_ab.txt
