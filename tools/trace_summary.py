#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV (too large to keep) into a small JSON: per-kernel-family time,
device busy fraction and the distribution of idle gaps between consecutive kernels of the LAST SAMPLING PASS.
The pass is cut by kernel names (round 4): from the first k_timestep_embed behind the
(ddim_steps + 1)-th last k_ddim_step (or the start of the trace) to the last k_ddim_step -- the hint block and the context
projections of that pass's first step included, the first-stage decodes behind it excluded.
Usage: python tools/trace_summary.py <..._kernel_trace.csv> <out.json> [ddim_steps=50]"""
import csv
import json
import sys

import numpy as np


def family(name):
    if 'igemm' in name or 'splitk_reduce' in name:
        return 'igemm'
    if 'attn' in name:
        return 'attention'
    if 'gn_' in name or 'ln_kernel' in name or 'row_stats' in name:
        return 'norm'
    return 'other'


def main():
    path, out = sys.argv[1:3]
    nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    steps = [i for i, r in enumerate(rows) if 'k_ddim_step' in r[2]]
    if len(steps) < nsteps:
        raise SystemExit(f'trace holds {len(steps)} k_ddim_step launches, need {nsteps}')
    last = steps[-1]
    prev_end = steps[-nsteps - 1] if len(steps) > nsteps else -1       # last sampler update of the pass before this one
    # the pass's own set-up work (hint block, context K / V projections) sits between prev_end and its first k_timestep_embed:
    # count it, as bench.py's timed region does
    first = prev_end + 1
    rows = rows[first:last + 1]
    st = np.array([r[0] for r in rows], dtype=np.int64)
    en = np.array([r[1] for r in rows], dtype=np.int64)
    dur = en - st
    span = int(en.max() - st.min())
    gaps = st[1:] - np.maximum.accumulate(en)[:-1]
    gaps = np.clip(gaps, 0, None)
    fam = {}
    for (s, e, n), d in zip(rows, dur):
        f = fam.setdefault(family(n), [0, 0])
        f[0] += 1
        f[1] += int(d)
    big = [int(i) for i in np.nonzero(gaps > 20000)[0][:40]]
    big_list = [{'gap_us': float(gaps[i]) / 1e3, 'after': rows[i][2][:60], 'before': rows[i + 1][2][:60],
                 'kernel_index': i} for i in big]
    # who sits on either side of the small gaps (2 us .. 20 us)?  total gap time by the kernel that starts after the gap
    def short(n):
        n = n.replace('(anonymous namespace)::', '').replace('void ', '')
        return n.split('(')[0][:70]
    by_next, by_prev = {}, {}
    for i in np.nonzero((gaps > 2000) & (gaps <= 20000))[0]:
        for d, k in ((by_next, short(rows[i + 1][2])), (by_prev, short(rows[i][2]))):
            v = d.setdefault(k, [0, 0])
            v[0] += 1
            v[1] += int(gaps[i])
    top = lambda d: [{'kernel': k, 'gaps': v[0], 'ms': v[1] / 1e6} for k, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:12]]
    res = {
        'window': f'last sampling pass: kernels {first}..{last} of the trace ({nsteps} k_ddim_step launches; decode excluded)',
        'kernels': len(rows), 'span_ms': span / 1e6, 'kernel_sum_ms': float(dur.sum()) / 1e6,
        'busy_fraction': float(dur.sum()) / span,
        'gap_total_ms': float(gaps.sum()) / 1e6, 'gap_mean_us': float(gaps.mean()) / 1e3,
        'gap_median_us': float(np.median(gaps)) / 1e3, 'gap_p90_us': float(np.percentile(gaps, 90)) / 1e3,
        'gap_p99_us': float(np.percentile(gaps, 99)) / 1e3, 'gaps_over_20us': int((gaps > 20000).sum()),
        'gap_ms_in_gaps_over_20us': float(gaps[gaps > 20000].sum()) / 1e6,
        'big_gaps': big_list,
        'small_gaps_by_following_kernel': top(by_next), 'small_gaps_by_preceding_kernel': top(by_prev),
        'families': {k: {'launches': v[0], 'ms': v[1] / 1e6, 'avg_us': v[1] / v[0] / 1e3} for k, v in fam.items()},
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
