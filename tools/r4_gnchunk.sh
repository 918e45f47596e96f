#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
for ch in 64 128 256 512; do
  echo "== FGDM_GN_CHUNK=$ch"
  FGDM_GN_CHUNK=$ch timeout -k 10 200 python tools/bench_norm.py 2>/dev/null | grep "HW4096\|B2 "
done | tee $OUT/gnchunk.txt
for r in 1 2; do for ch in 64 256; do
  FGDM_GN_CHUNK=$ch timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('gn chunk $ch: %.3f img/s  norm est %.1f ms' % (d['value'], d['kernel_time_ms_est']['norm']))" || exit 1
done; done | tee -a $OUT/gnchunk.txt
