#!/bin/bash
# Round 4 closing measurements on ONE box: smoke, default bench line, per-shape table, rocprofv3 profiles.
OUT=gpurun_out; mkdir -p $OUT/r4
timeout -k 10 600 python __graft_entry__.py --smoke > $OUT/r4/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/r4/smoke.log
timeout -k 10 900 python bench.py > $OUT/r04_bench_c3_default.json 2> $OUT/r4/bench_default.err; echo "bench rc=$?"; cut -c1-400 $OUT/r04_bench_c3_default.json
FGDM_PROF_DUMP=$OUT/r04_per_shape_times.tsv timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage > /dev/null 2>&1
bash tools/profile_round.sh r04 > $OUT/r4/profile_round.log 2>&1; echo "profile rc=$?"; tail -16 $OUT/r4/profile_round.log
