#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV (too large to keep) into a small JSON: per-kernel-family time,
device busy fraction and the distribution of idle gaps between consecutive kernels of the last sampling pass.
Usage: python tools/trace_summary.py <..._kernel_trace.csv> <out.json> [fraction_of_trace_to_keep=0.45]"""
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
    keep = float(sys.argv[3]) if len(sys.argv) > 3 else 0.45
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    rows = rows[int(len(rows) * (1.0 - keep)):]          # tail of the trace = inside the timed sampling pass
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
    res = {
        'kernels': len(rows), 'span_ms': span / 1e6, 'kernel_sum_ms': float(dur.sum()) / 1e6,
        'busy_fraction': float(dur.sum()) / span,
        'gap_total_ms': float(gaps.sum()) / 1e6, 'gap_mean_us': float(gaps.mean()) / 1e3,
        'gap_median_us': float(np.median(gaps)) / 1e3, 'gap_p90_us': float(np.percentile(gaps, 90)) / 1e3,
        'gap_p99_us': float(np.percentile(gaps, 99)) / 1e3, 'gaps_over_20us': int((gaps > 20000).sum()),
        'gap_ms_in_gaps_over_20us': float(gaps[gaps > 20000].sum()) / 1e6,
        'big_gaps': big_list,
        'families': {k: {'launches': v[0], 'ms': v[1] / 1e6, 'avg_us': v[1] / v[0] / 1e3} for k, v in fam.items()},
    }
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res))


if __name__ == '__main__':
    main()
