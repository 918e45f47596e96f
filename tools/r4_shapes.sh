#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
FGDM_PROF_DUMP=$OUT/shapes_cn0_p8.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --controlnets 0 --prompts 8 --profile-stride 1 > $OUT/bench_cn0_p8_stride1.json 2>/dev/null
FGDM_PROF_DUMP=$OUT/shapes_c3.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage --profile-stride 1 > $OUT/bench_c3_stride1.json 2>/dev/null
wc -l $OUT/shapes_*.tsv
