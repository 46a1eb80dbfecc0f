#!/usr/bin/env python3
"""Print a merged GPU timeline (kernels + memory copies) of a few steady-state steps from a
`rocprofv3 --kernel-trace --memory-copy-trace --output-format csv` run.  usage: timeline.py <dir> [n_events]"""
import csv, glob, os, sys
d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 80
ev = []
for f in glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'K q%s ' % r.get('Queue_Id', '?') + r['Kernel_Name'][:48]))
for f in glob.glob(os.path.join(d, '**', '*memory_copy_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'C ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
ev.sort()
marks = [i for i, e in enumerate(ev) if 'deconv_last' in e[2]]
i0 = marks[len(marks) // 2] + 1
t0 = ev[i0][0]
for s, e, name in ev[i0:i0 + n]:
    print(f'{(s - t0) / 1e6:9.3f} ms  +{(e - s) / 1e3:9.1f} us  {name}')
