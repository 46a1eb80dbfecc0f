#!/usr/bin/env python3
"""GPU busy/idle accounting from a `rocprofv3 --kernel-trace --output-format csv` run of bench.py.

usage: gpu_idle.py <dir>   -> span of the kernel timeline, union of kernel intervals (busy), idle gaps by size,
and the mean duration per kernel name inside the steady state (middle 60 % of the timeline)."""
import collections, csv, glob, os, sys

kt = glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(kt))]
rows.sort()
# steady state = between the 25 % and 75 % occurrences of the last synthesis kernel (bench's timed loop dominates)
marks = ([r for r in rows if 'deconv_last' in r[2]] or [r for r in rows if 'pmap_gather' in r[2]] or
         [r for r in rows if 'conv_first' in r[2]] or rows)
lo, hi = marks[len(marks) // 4][0], marks[(3 * len(marks)) // 4][1]
mid = [r for r in rows if r[0] >= lo and r[1] <= hi]
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e, _ in mid:
    if cur_e is None:
        cur_s, cur_e = s, e
    elif s <= cur_e:
        cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
busy += cur_e - cur_s
span = mid[-1][1] - mid[0][0]
print(f'steady-state window {span / 1e6:.2f} ms: busy {busy / 1e6:.2f} ms ({busy / span:.3f}), idle {1 - busy / span:.3f}')
hist = collections.Counter()
for g, _ in gaps:
    hist['<5us' if g < 5e3 else '<20us' if g < 2e4 else '<100us' if g < 1e5 else '<1ms' if g < 1e6 else '>=1ms'] += g
print('idle by gap size (ms):', {k: round(v / 1e6, 3) for k, v in hist.items()})
acc = collections.defaultdict(list)
for s, e, k in mid:
    acc[k[:72]].append(e - s)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f'{sum(v) / 1e6:9.2f} ms  n={len(v):4d}  mean {sum(v) / len(v) / 1e3:9.1f} us  {k}')
