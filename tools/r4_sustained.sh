#!/bin/bash
# the driver's command (20 timed steps after 5 warm-up steps: ~50 s of sustained load) and the 8-prompt configurations on the same box
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_sustained.json 2> $OUT/bench_sustained.err || exit 1
python - <<'PY'
import json
d = json.load(open('gpurun_out/r4/bench_sustained.json'))
print('sustained (20 steps): %.3f img/s  %.1f ms/step  igemm %.1f TF/s (%.3f)  whole path %.3f  timed launches %d' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['whole_path_mfma_frac'], d['roofline']['timed_launches']))
PY
bash tools/r4_cfgs.sh
