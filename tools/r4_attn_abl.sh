#!/bin/bash
# Round 4: ablations of the two-strand attention kernel (what each part of its stream costs), ONE gpurun call.
OUT=gpurun_out/r4; mkdir -p $OUT; : > $OUT/attn_abl.txt
for ds in 1 0; do
  for a in 0 1 2 3 4 7 16 32 48 51; do
    FGDM_BENCH_DATA_SCALE=$ds FGDM_ATTN_DQ=1 FGDM_ATTN_ABL=$a timeout -k 10 120 python tools/bench_attention.py --only 0 --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/scale=$ds ABL=$a /" | tee -a $OUT/attn_abl.txt
  done
done
