#!/bin/bash
# Profiles of one round, to be run on the GPU box inside ONE gpurun call (tag = e.g. r02):
#   bash tools/profile_round.sh r02
# 1. rocprofv3 --kernel-trace --stats of the default bench command            -> gpurun_out/<tag>_kernel_stats.csv (+ bench JSON)
# 2. PMC FETCH_SIZE and WRITE_SIZE in SEPARATE passes (MI355X guide, HBM)     -> gpurun_out/<tag>_pmc_traffic.json
# 3. SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES / GRBM_GUI_ACTIVE in one more pass       -> gpurun_out/<tag>_mfma_busy.txt
# Copy the summaries into profiles/ afterwards (gpurun_out/ is scratch).
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline \
    > gpurun_out/${TAG}_bench_under_rocprof.json 2> gpurun_out/${TAG}_prof.log || exit 1
cp $(find gpurun_out/${TAG}_prof -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv
python3 tools/trace_summary.py $(find gpurun_out/${TAG}_prof -name '*kernel_trace.csv' | head -1) gpurun_out/${TAG}_trace_summary.json >/dev/null 2>&1
rm -rf gpurun_out/${TAG}_prof
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/${TAG}_pmc_$c -o p -- python3 bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-first-stage \
      > /dev/null 2>> gpurun_out/${TAG}_prof.log || exit 1
done
python3 tools/pmc_summary.py $(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1) \
    $(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1) gpurun_out/${TAG}_pmc_traffic.json > /dev/null
rm -rf gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/${TAG}_pmc_mfma -o p -- python3 bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-first-stage \
    > /dev/null 2>> gpurun_out/${TAG}_prof.log || exit 1
python3 tools/pmc_mfma.py $(find gpurun_out/${TAG}_pmc_mfma -name '*counter_collection.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv > gpurun_out/${TAG}_mfma_busy.txt
rm -rf gpurun_out/${TAG}_pmc_mfma
head -12 gpurun_out/${TAG}_kernel_stats.csv | cut -c1-160
cat gpurun_out/${TAG}_mfma_busy.txt
