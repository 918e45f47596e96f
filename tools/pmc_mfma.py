#!/usr/bin/env python3
"""Matrix-pipe utilisation per kernel from ONE rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES
GRBM_GUI_ACTIVE; no trace options in that pass) of the bench command: the table VERDICT r3 asked to be tracked.

  cycles     = GRBM_GUI_ACTIVE / 8            (rocprofv3 reports the sum over the 8 XCDs: MI355X guide, DVFS note)
  clock      = cycles / duration              (duration: the counter pass has no timestamps, so the average of the same kernel
                                               in the --kernel-trace --stats run of the same command; reads high for launches
                                               much shorter than 0.3 ms, as the guide says)
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles)     (the counter is 32 per v_mfma_f32_32x32x16 wave-instruction,
                                               16 per 16x16x32: pipe cycles summed over the SIMDs)
  cu_busy    = SQ_BUSY_CU_CYCLES / (256 CUs x cycles)  (summed over the CUs; some builds of the counter report per-SE sums: the
                                               column is printed raw-normalised and is only comparable between rows)
  peak_frac  = mfma_busy x clock / 2.4 GHz    (fraction of the 2.5 PF dense peak the kernel's EXECUTED MFMAs reach)
Usage: python tools/pmc_mfma.py <counter_collection.csv> <kernel_stats.csv> [top=12]"""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r'\(anonymous namespace\)::', '', name)
    name = re.sub(r'^void ', '', name)
    return name[:86]


def main():
    pmc, stats = sys.argv[1:3]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    dur = {r['Name']: (float(r['AverageNs']), int(r['Calls'])) for r in csv.DictReader(open(stats))}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(pmc)):
        k = r['Kernel_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        disp[k].add(r.get('Dispatch_Id'))
    rows = []
    for k, c in agg.items():
        if k not in dur:
            continue
        n = len(disp[k])
        avg_ns, calls = dur[k]
        cyc = c.get('GRBM_GUI_ACTIVE', 0.0) / n / 8.0
        if cyc <= 0:
            continue
        busy = c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / n
        cub = c.get('SQ_BUSY_CU_CYCLES', 0.0) / n
        clock = cyc / avg_ns                      # GHz
        mf = busy / (1024.0 * cyc)
        rows.append((avg_ns * calls, short(k), calls, avg_ns / 1e3, clock, mf, cub / (256.0 * cyc), mf * clock / 2.4))
    rows.sort(reverse=True)
    print(f"{'kernel (by total time in the traced bench run)':86s} {'calls':>6s} {'avg us':>8s} {'clock GHz':>9s} {'mfma_busy':>9s} {'cu_busy':>8s} {'peak_frac':>9s}")
    for _, k, calls, us, clk, mf, cb, pf in rows[:top]:
        print(f'{k:86s} {calls:6d} {us:8.1f} {clk:9.2f} {mf:9.3f} {cb:8.2f} {pf:9.3f}')


if __name__ == '__main__':
    main()
