#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
for v in 0 1; do
  if [ $v = 1 ]; then export FGDM_DEBUG_VT_ROWMAJOR=1; fi
  FGDM_PROF_DUMP=$OUT/vt_probe_$v.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --ddim-steps 10 --no-cpu-baseline --no-first-stage --profile-stride 1 > /dev/null 2>&1
  echo "== row-major V (timing only) = $v"
  grep -E "N960 K320|N1920 K640|N3840 K1280" $OUT/vt_probe_$v.tsv | awk -F'\t' '{printf "%-56s %5d %9.3f ms %8.1f us\n",$1,$2,$3,$3/$2*1000}'
done
