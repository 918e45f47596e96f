#!/bin/bash
# Round 4: the two-strand attention kernels against the shipped one, inside ONE gpurun call (boxes differ by several percent).
# usage: bash tools/r4_attn_ab.sh "<DQ values to test>" "<DQ values to bench>" "<ABL list for DQ in $4>" <DQ for ablations>
OUT=gpurun_out/r4; mkdir -p $OUT; : > $OUT/attn_bench.txt
for v in $1; do
  FGDM_ATTN_DQ=$v timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "attention" > $OUT/attn_tests_dq$v.log 2>&1
  echo "tests DQ=$v rc=$?"; tail -3 $OUT/attn_tests_dq$v.log
done
for v in $2; do
  for i in 0 7; do
    FGDM_ATTN_DQ=$v timeout -k 10 120 python tools/bench_attention.py --only $i --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/DQ=$v /" | tee -a $OUT/attn_bench.txt
  done
  FGDM_BENCH_DATA_SCALE=0 FGDM_ATTN_DQ=$v timeout -k 10 120 python tools/bench_attention.py --only 0 --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/zeros DQ=$v /" | tee -a $OUT/attn_bench.txt
done
for ds in 1 0; do
  for a in $3; do
    FGDM_BENCH_DATA_SCALE=$ds FGDM_ATTN_DQ=$4 FGDM_ATTN_ABL=$a timeout -k 10 120 python tools/bench_attention.py --only 0 --iters 20 2>&1 | grep -v amdgpu.ids | sed "s/^/scale=$ds DQ=$4 ABL=$a /" | tee -a $OUT/attn_bench.txt
  done
done
