#!/bin/bash
# Round 4: the halo-tile convolution loop against the per-tap loop, ONE gpurun call.
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv" > $OUT/halo_tests.log 2>&1; echo "conv tests rc=$?"; tail -4 $OUT/halo_tests.log
: > $OUT/halo_bench.txt
for h in 0 1 0 1; do
  FGDM_IGEMM_HALO=$h timeout -k 10 300 python tools/bench_igemm.py --iters 20 --cfgs 0 --shapes "conv" 2>&1 | grep -v amdgpu.ids | sed "s/^/HALO=$h /" | tee -a $OUT/halo_bench.txt
done
