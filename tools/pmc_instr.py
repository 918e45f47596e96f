#!/usr/bin/env python3
"""Instruction mix per kernel from one rocprofv3 --pmc pass (e.g. SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES) next to
the kernel durations of a --kernel-trace --stats run: which kernels are bound by their vector-ALU instruction stream?
`valu_busy` = wave-instructions x 4 cycles / 1024 SIMDs / 2.1 GHz / duration (transcendentals and fp64 issue slower, so it
is a lower bound).  Usage: python tools/pmc_instr.py <counter_collection.csv> <kernel_stats.csv>"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:78]


def main():
    pmc, stats = sys.argv[1:3]
    dur = {r['Name']: float(r['AverageNs']) for r in csv.DictReader(open(stats))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(pmc)):
        k = r['Kernel_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r.get('Dispatch_Id'), k)
        if key not in seen:
            seen.add(key)
            n[k] += 1
    rows = []
    for k, c in agg.items():
        d = dur.get(k)
        if not d or not n[k]:
            continue
        valu = c.get('SQ_INSTS_VALU', 0.0) / n[k]
        busy = valu * 4 / 1024 / 2.1e9 / (d * 1e-9)
        rows.append((d * n[k], short(k), n[k], d / 1e3, valu, c.get('SQ_INSTS_SALU', 0.0) / n[k], c.get('SQ_INSTS_LDS', 0.0) / n[k],
                     c.get('SQ_WAVES', 0.0) / n[k], busy))
    rows.sort(reverse=True)
    print(f"{'kernel':78s} {'n':>5s} {'us':>8s} {'VALU/launch':>12s} {'SALU':>11s} {'LDS':>11s} {'waves':>8s} {'valu_busy':>9s}")
    for _, k, m, us, v, s, l, w, b in rows[:30]:
        print(f'{k:78s} {m:5d} {us:8.1f} {v:12.0f} {s:11.0f} {l:11.0f} {w:8.0f} {b:9.2f}')


if __name__ == '__main__':
    main()
