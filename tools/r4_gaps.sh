#!/bin/bash
# kernel trace of the default bench command -> gap attribution (tools/trace_summary.py)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r4; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/gaps_prof -o bench -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-first-stage > $OUT/gaps_bench.json 2> $OUT/gaps_prof.log || exit 1
python3 tools/trace_summary.py $(find $OUT/gaps_prof -name '*kernel_trace.csv' | head -1) $OUT/gaps_summary.json > /dev/null || exit 1
rm -rf $OUT/gaps_prof
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r4/gaps_summary.json'))
print('span', d['span_ms'], 'busy', d['busy_fraction'], 'gap total', d['gap_total_ms'])
for k in ('small_gaps_by_following_kernel', 'small_gaps_by_preceding_kernel'):
    print(k)
    for r in d[k]: print('  %-72s %6d %8.3f ms' % (r['kernel'], r['gaps'], r['ms']))
PY
