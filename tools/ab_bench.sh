# A/B two settings of one environment variable inside ONE gpurun call (boxes differ by several percent):
#   bash tools/ab_bench.sh VAR A B [rounds]
mkdir -p gpurun_out
VAR=$1; A=$2; B=$3; R=${4:-2}
for i in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
fs = d.get('first_stage_decode') or {}
print('$VAR=$v: %.3f img/s  igemm %.0f TF/s  attn %.0f TF/s  kernel ms %s  decode %.2f ms/img' % (d['value'], d['roofline']['achieved'], d['attention_tflops'], d['kernel_time_ms_est'], fs.get('ms_per_image', float('nan'))))" || exit 1
  done
done
