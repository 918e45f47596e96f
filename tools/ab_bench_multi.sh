# A/B several settings of one environment variable inside ONE gpurun call (boxes differ by several percent):
#   bash tools/ab_bench_multi.sh VAR "v1 v2 v3" [rounds]
mkdir -p gpurun_out
VAR=$1; VALS=$2; R=${3:-2}
for i in $(seq 1 $R); do
  for v in $VALS; do
    env $VAR=$v timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('$VAR=$v: %.3f img/s  igemm %.0f TF/s  attn %.0f TF/s  kernel ms %s' % (d['value'], d['roofline']['achieved'], d['attention_tflops'], d['kernel_time_ms_est']))" || exit 1
  done
done
