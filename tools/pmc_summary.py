#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE collected in SEPARATE runs, as the MI355X guide
prescribes) into HBM bytes per launch for each kernel family.

gfx950 corrections (MI355X_MICROARCH.md, section HBM): both counters are in KiB; FETCH_SIZE reports exactly half of the
bytes of wide coalesced reads -> doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def family(name):
    if 'igemm' in name or 'splitk_reduce' in name:
        return 'igemm'
    if 'attn' in name:
        return 'attention'
    if 'gn_' in name or 'ln_kernel' in name or 'row_stats' in name:
        return 'norm'
    return 'other'


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        f = family(r['Kernel_Name'])
        agg[f][0] += 1
        agg[f][1] += float(r['Counter_Value'])
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, 'FETCH_SIZE'), load(write, 'WRITE_SIZE')
    res = {}
    for fam in sorted(set(f) | set(w)):
        n = f[fam][0] or w[fam][0]
        rd = 2.0 * f[fam][1] * 1024.0          # x2: gfx950 FETCH_SIZE under-count of wide coalesced reads
        wr = w[fam][1] * 1024.0
        res[fam] = {'launches': n, 'hbm_read_bytes_per_launch': rd / max(n, 1),
                    'hbm_write_bytes_per_launch': wr / max(n, 1),
                    'hbm_bytes_per_launch': (rd + wr) / max(n, 1)}
    res['_note'] = ('rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over '
                    '`bench.py --steps 1 --warmup 0 --ddim-steps 2`; FETCH_SIZE x 2 x 1024, WRITE_SIZE x 1024')
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == '__main__':
    main()
