#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv or attention" > $OUT/step3_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/step3_tests.log
for v in 0 1; do
  FGDM_ATTN_DQ80=$v timeout -k 10 120 python tools/bench_attention.py --only 2 --iters 30 2>&1 | grep -v amdgpu.ids | sed "s/^/DQ80=$v /" | tee -a $OUT/attn80_bench2.txt
done
bash tools/ab_bench_multi.sh FGDM_IGEMM_HALO "0 1" 2 > $OUT/ab_halo_e2e2.txt 2>&1; cat $OUT/ab_halo_e2e2.txt
bash tools/ab_bench_multi.sh FGDM_ATTN_DQ80 "0 1" 1 > $OUT/ab_dq80_e2e.txt 2>&1; cat $OUT/ab_dq80_e2e.txt
