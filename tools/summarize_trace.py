#!/usr/bin/env python3
"""Per-(kernel, grid) summary of a rocprofv3 --kernel-trace CSV (the same kernel template serves
several layers; the grid size tells them apart).  usage: summarize_trace.py trace.csv [skip_first_n_calls]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
groups = collections.OrderedDict()
for r in rows:
    name = r['Kernel_Name']
    grid = int(r['Grid_Size_X']) if 'Grid_Size_X' in r else int(r.get('Grid_Size', 0))
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    groups.setdefault((name, grid), []).append(dur)
print('| kernel | grid (threads) | calls | avg us | min us | max us |')
print('|---|---|---|---|---|---|')
for (name, grid), d in groups.items():
    d = d[skip:] if len(d) > skip else d
    print(f'| `{name}` | {grid} | {len(d)} | {sum(d)/len(d):.1f} | {min(d):.1f} | {max(d):.1f} |')
