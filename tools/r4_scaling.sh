#!/bin/bash
# how do the short-K layers' times scale with the number of tiles per CU (fixed cost per round of tiles)?
OUT=gpurun_out/r4; mkdir -p $OUT; : > $OUT/tile_scaling.txt
for b in 8 16 32 64 128; do
  timeout -k 10 300 python tools/bench_igemm.py --iters 30 --cfgs 0 --batch $b --shapes "L0 lin 320->320,L0 geglu,L0 lin 1280->320,L1 lin 640->640,L1 geglu,L0 conv 320->320" 2>&1 | grep -v "amdgpu.ids\|shape" | sed "s/^/B=$b /" | tee -a $OUT/tile_scaling.txt
done
