#!/usr/bin/env python3
"""Per-launch HBM traffic of every kernel from two `rocprofv3 --pmc` passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

usage: hbm_traffic.py <fetch dir> <write dir>  -> JSON on stdout:
  {"<kernel short name>@<grid threads>": {fetch_kb_raw, fetch_mb_x2, write_mb, hbm_mb}}
FETCH_SIZE is in KiB... as reported by rocprofv3 (KB); it is doubled per MI355X_MICROARCH.md (gfx950 tallies the
128-byte requests of wide coalesced reads as 64 bytes).  Check: an elementwise kernel (dequantize_kernel) must
come out at its exact byte count."""
import collections, csv, glob, json, os, re, sys


def per_kernel(d, counter):
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    acc, n = collections.defaultdict(float), collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        name = re.sub(r'^void ', '', r['Kernel_Name'])
        name = re.sub(r'\(.*$', '', name).replace('cae::', '')
        key = f"{name}@{r['Grid_Size']}"
        acc[key] += float(r['Counter_Value'])
        n[key].add(r['Dispatch_Id'])
    return {k: acc[k] / len(n[k]) for k in acc}


fetch = per_kernel(sys.argv[1], 'FETCH_SIZE')
write = per_kernel(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in fetch:
    if not any(t in k for t in ('conv', 'quantize', 'nchw', 'tile_sse', 'u64', 'likelihood', 'pmap')):
        continue
    fx2 = 2 * fetch[k] * 1024 / 1e6
    w = write.get(k, 0.0) * 1024 / 1e6
    out[k] = dict(fetch_kb_raw=fetch[k], fetch_mb_x2=fx2, write_mb=w, hbm_mb=fx2 + w)
print(json.dumps(out, indent=1))
