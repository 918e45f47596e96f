#!/bin/bash
# Per-shape kernel times of the 8-prompt configurations (BASELINE configs[1], [3], [4] per GPU): where do they lose against 16 prompts?
OUT=gpurun_out/r4; mkdir -p $OUT
for cn in 0 2 3; do
  FGDM_PROF_DUMP=$OUT/shapes_cn${cn}_p8.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --prompts 8 --controlnets $cn --no-cpu-baseline --no-first-stage --profile-stride 1 \
      > $OUT/shapes_cn${cn}_p8.json 2> $OUT/shapes_cn${cn}_p8.err || exit 1
  cut -c1-200 $OUT/shapes_cn${cn}_p8.json
done
