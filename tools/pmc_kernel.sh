#!/bin/bash
# Hardware counters of ONE micro-benchmarked GEMM shape (separate --pmc passes, no trace options):
#   bash tools/pmc_kernel.sh TAG "shape substring" CFG   -> gpurun_out/<TAG>_pmc.txt (average per dispatch of the timed kernel)
#   bash tools/pmc_kernel.sh TAG attn                     -> the same for tools/bench_attention.py (first case)
TAG=${1:-pmc}; SHAPE=${2:-L0 conv 320}; CFG=${3:-4}
if [ "$SHAPE" = attn ]; then CMD=(python3 tools/bench_attention.py --iters 3 --only 0); else CMD=(python3 tools/bench_igemm.py --iters 3 --cfgs $CFG --shapes "$SHAPE"); fi
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p "$(dirname gpurun_out/${TAG}_x)"
OUT=gpurun_out/${TAG}_pmc.txt; : > $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_COEXEC_CYCLES SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_p$i
  rocprofv3 --pmc $set --output-format csv -d gpurun_out/${TAG}_p$i -o p -- "${CMD[@]}" > /dev/null 2>> gpurun_out/${TAG}_pmc.log
  f=$(find gpurun_out/${TAG}_p$i -name '*counter_collection.csv' | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f >> $OUT; else echo "pass $i failed: $set" >> $OUT; fi
  rm -rf gpurun_out/${TAG}_p$i
done
cat $OUT
