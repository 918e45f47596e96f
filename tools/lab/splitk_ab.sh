mkdir -p gpurun_out/r3
for cfg in "4 2" "8 2" "8 0" "4 0" "2 0" "2 2" "16 0"; do set -- $cfg; echo "factor $1 tile $2"; FGDM_SPLITK_FACTOR=$1 FGDM_SPLITK_TILE=$2 timeout -k 10 120 python tools/bench_igemm.py --shapes "L3 conv,L3 lin 5120" --cfgs 0 2>&1 | grep -v amdgpu.ids | tail -3; done
