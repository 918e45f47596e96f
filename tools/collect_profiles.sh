#!/bin/bash
# Run HERE (not on the GPU box) after `gpurun -- bash tools/profile_round.sh TAG`: copies the round's summaries from the scratch
# directory gpurun_out/ into the tracked profiles/ and stamps the JSON files with the commit the profiled tree was built from
# (the GPU box has no .git), so that bench.py can label what it quotes from them ("source": "profiles/<file> @ <commit>").
TAG=${1:?tag, e.g. r04}
C=$(git rev-parse --short HEAD)$(git diff --quiet || echo +dirty)
for f in kernel_stats.csv trace_summary.json pmc_traffic.json mfma_busy.txt bench_under_rocprof.json; do
  [ -f gpurun_out/${TAG}_$f ] && cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
python3 - "$TAG" "$C" <<'PY'
import json, sys
tag, commit = sys.argv[1:3]
for name in ('trace_summary', 'pmc_traffic'):
    p = f'profiles/{tag}_{name}.json'
    try:
        d = json.load(open(p))
    except Exception:
        continue
    d['_source_commit'] = commit
    json.dump(d, open(p, 'w'), indent=1)
    print('stamped', p, commit)
PY
