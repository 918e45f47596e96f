#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "conv" > $OUT/step4_tests.log 2>&1; echo "conv tests rc=$?"; tail -3 $OUT/step4_tests.log
: > $OUT/halo_bench2.txt
for h in 0 1 0 1; do
  FGDM_IGEMM_HALO=$h timeout -k 10 300 python tools/bench_igemm.py --iters 20 --cfgs 0 --shapes "L2 conv,L3 conv" 2>&1 | grep -v "amdgpu.ids\|shape" | sed "s/^/HALO=$h /" | tee -a $OUT/halo_bench2.txt
done
bash tools/ab_bench_multi.sh FGDM_IGEMM_HALO "0 1" 2 > $OUT/ab_halo_e2e3.txt 2>&1; cat $OUT/ab_halo_e2e3.txt
bash tools/r4_cfgs.sh
