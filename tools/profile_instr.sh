#!/bin/bash
# Instruction counters per kernel (one --pmc pass, no trace options) -> gpurun_out/<tag>_instr.txt
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --output-format csv -d gpurun_out/${TAG}_pmc_instr -o p -- python3 bench.py --steps 1 --warmup 0 --ddim-steps 2 --no-cpu-baseline --no-first-stage \
    > /dev/null 2> gpurun_out/${TAG}_instr.log || exit 1
STATS=profiles/${TAG}_kernel_stats.csv; [ -f "$STATS" ] || STATS=gpurun_out/${TAG}_kernel_stats.csv
[ -f "$STATS" ] || { echo "no kernel statistics for tag ${TAG}: run tools/profile_round.sh ${TAG} first" >&2; exit 1; }
python3 tools/pmc_instr.py $(find gpurun_out/${TAG}_pmc_instr -name '*counter_collection.csv' | head -1) "$STATS" > gpurun_out/${TAG}_instr.txt
rm -rf gpurun_out/${TAG}_pmc_instr
cat gpurun_out/${TAG}_instr.txt
