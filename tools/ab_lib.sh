# A/B two BUILDS of the library inside ONE gpurun call (boxes differ by several percent):
#   bash tools/ab_lib.sh fgdm_amd/libfgdm_hip_A.so [rounds]
# alternates the given library (A) with the in-tree one (B) under the default bench.  The in-tree file is never overwritten
# (ADVICE r3): each run loads its build through FGDM_LIB, which fgdm_amd/_lib.py honours.
A=$1; R=${2:-2}
[ -f "$A" ] || { echo "no such library: $A" >&2; exit 1; }
for i in $(seq 1 "$R"); do
  for v in A B; do
    if [ $v = A ]; then L="$A"; else L=fgdm_amd/libfgdm_hip.so; fi
    FGDM_LIB="$L" timeout -k 10 300 python bench.py --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('build $v: %.3f img/s  igemm %.0f TF/s  kernel ms %s' % (d['value'], d['roofline']['achieved'], d['kernel_time_ms_est']))"
  done
done
