#!/bin/bash
OUT=gpurun_out/r4; mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "layernorm or linear or geglu" > $OUT/persist_tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/persist_tests.log
: > $OUT/persist_bench.txt
for h in 0 1 0 1; do
  FGDM_IGEMM_PERSIST=$h timeout -k 10 300 python tools/bench_igemm.py --iters 20 --cfgs 0 --shapes "geglu" 2>&1 | grep -v "amdgpu.ids\|shape" | sed "s/^/PERSIST=$h /" | tee -a $OUT/persist_bench.txt
done
bash tools/ab_bench_multi.sh FGDM_IGEMM_PERSIST "0 1" 2 > $OUT/ab_persist_e2e.txt 2>&1; cat $OUT/ab_persist_e2e.txt
