#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_full_size_multi.py -q -x -m gpu -k "grouped or several or full_size_with" --durations=8 > $OUT/step2_tests.log 2>&1; echo "tests rc=$?"; tail -14 $OUT/step2_tests.log
FGDM_ATTN_DQ80=1 timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "attention" > $OUT/step2_attn80.log 2>&1; echo "attn80 tests rc=$?"; tail -3 $OUT/step2_attn80.log
for v in 0 1 0 1; do
  FGDM_ATTN_DQ80=$v timeout -k 10 120 python tools/bench_attention.py --only 2 --iters 30 2>&1 | grep -v amdgpu.ids | sed "s/^/DQ80=$v /" | tee -a $OUT/attn80_bench.txt
done
