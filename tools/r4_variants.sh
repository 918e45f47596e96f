#!/bin/bash
# per-shape, per-epilogue-variant kernel times of the default workload (every launch bracketed)
OUT=gpurun_out/r4; mkdir -p $OUT
FGDM_PROF_DUMP=$OUT/shapes_variants.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-first-stage --profile-stride 1 > $OUT/shapes_variants.json 2>/dev/null || exit 1
grep -E "N320 K320|N640 K640|N1280 K1280 |N960 K320|N320 K1280" $OUT/shapes_variants.tsv | sort -t$'\t' -k3 -n -r | awk -F'\t' '{printf "%-56s %5d %9.3f ms %8.1f us %7.0f TF/s\n",$1,$2,$3,$3/$2*1000,$4/$3/1e9}'
