#!/usr/bin/env python3
"""Average counter values per dispatch for the igemm kernels of one rocprofv3 --pmc pass (tools/pmc_kernel.sh)."""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    if 'igemm' not in k and 'attn' not in k:
        continue
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    disp[k].add(r.get('Dispatch_Id'))
for k, c in agg.items():
    n = len(disp[k])
    print(k[:150], 'dispatches', n)
    for name, v in sorted(c.items()):
        print(f'    {name:32s} {v / n:16.0f}')
