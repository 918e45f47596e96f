#!/bin/bash
# Round 4: the two-strand attention kernel against the shipped one, inside ONE gpurun call (boxes differ by several percent).
OUT=gpurun_out/r4; mkdir -p $OUT
./tools/lab/permlane_probe.bin > $OUT/permlane_probe.txt 2>&1; cat $OUT/permlane_probe.txt
for v in 1 2; do
  FGDM_ATTN_DQ=$v timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "attention" > $OUT/attn_tests_dq$v.log 2>&1
  echo "tests DQ=$v rc=$?"; tail -3 $OUT/attn_tests_dq$v.log
done
for v in 0 1 2; do
  for i in 0 7; do
    FGDM_ATTN_DQ=$v timeout -k 10 120 python tools/bench_attention.py --only $i --iters 20 2>&1 | sed "s/^/DQ=$v /" | tee -a $OUT/attn_bench.txt
  done
done
for v in 1 2; do
  FGDM_BENCH_DATA_SCALE=0 FGDM_ATTN_DQ=$v timeout -k 10 120 python tools/bench_attention.py --only 0 --iters 20 2>&1 | sed "s/^/zeros DQ=$v /" | tee -a $OUT/attn_bench.txt
done
