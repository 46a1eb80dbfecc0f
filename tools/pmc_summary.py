#!/usr/bin/env python3
"""Per-kernel means of a `rocprofv3 --pmc ... --kernel-trace --output-format csv` run.

usage: pmc_summary.py <dir with *_counter_collection.csv and *_kernel_trace.csv>
Prints per kernel: launches, mean duration, every collected counter (mean per launch) and, when present,
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration and MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES)
(MI355X_MICROARCH.md, DVFS give-back)."""
import collections, csv, glob, os, sys

d = sys.argv[1]
cc = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
kt = glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True)[0]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (r['Kernel_Name'], int(r['End_Timestamp']) - int(r['Start_Timestamp']))
acc = collections.defaultdict(lambda: collections.defaultdict(float))
seen = collections.defaultdict(set)
for r in csv.DictReader(open(cc)):
    k = r['Kernel_Name'][:60] + ' grid=' + r['Grid_Size']
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    seen[k].add(r['Dispatch_Id'])
for k in sorted(acc, key=lambda k: -sum(dur[i][1] for i in seen[k] if i in dur)):
    n = len(seen[k])
    t = sum(dur[i][1] for i in seen[k] if i in dur) / max(1, n)
    c = {name: v / n for name, v in acc[k].items()}
    line = f'{k[:70]:70s} n={n:4d} {t / 1e6:8.3f} ms'
    if 'GRBM_GUI_ACTIVE' in c:
        line += f"  clock {c['GRBM_GUI_ACTIVE'] / 8 / t:5.2f} GHz"
    for name, v in sorted(c.items()):
        if name != 'GRBM_GUI_ACTIVE':
            line += f'  {name}={v:.4g}'
            if 'GRBM_GUI_ACTIVE' in c and name.startswith('SQ_'):
                line += f"({v / c['GRBM_GUI_ACTIVE']:.3f}/act)"
    print(line)
