#!/bin/bash
# HBM-side traffic (TCC FETCH_SIZE / WRITE_SIZE, separate passes) of single GEMM shapes, halo loop off / on:
# where does the family's 1.3x over the algorithmic bytes come from?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r4/pmc_shapes.txt; mkdir -p gpurun_out/r4; : > $OUT
for shape in "L0 conv 320->320" "L1 conv 640->640" "L2 conv 1280->1280" "L3 conv 1280->1280" "L0 lin 320->320" "L2 lin 1280->1280"; do
  for h in 0 1; do
    for c in FETCH_SIZE WRITE_SIZE; do
      rm -rf gpurun_out/r4/p_tmp
      FGDM_IGEMM_HALO=$h rocprofv3 --pmc $c --output-format csv -d gpurun_out/r4/p_tmp -o p -- python3 tools/bench_igemm.py --iters 3 --cfgs 0 --shapes "$shape" > /dev/null 2>> gpurun_out/r4/pmc_shapes.log
      f=$(find gpurun_out/r4/p_tmp -name '*counter_collection.csv' | head -1)
      echo "== $shape HALO=$h $c" >> $OUT
      [ -n "$f" ] && python3 tools/pmc_kernel.py $f >> $OUT
    done
  done
done
rm -rf gpurun_out/r4/p_tmp
cat $OUT
