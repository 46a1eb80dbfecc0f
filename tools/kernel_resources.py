#!/usr/bin/env python3
"""Summarise hipcc -Rpass-analysis=kernel-resource-usage for the library's kernels."""
import re, subprocess, sys, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, 'cnn_autoencoder_amd/csrc/cae_api.hip')
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(root, 'include'),
       '--offload-arch=gfx950', '-c', src, '-o', '/dev/null', '-Rpass-analysis=kernel-resource-usage']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'remark: (?:\s*)([A-Za-z ]+): (.*?) \[-Rpass', line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == 'Function Name':
        cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print('%-70s %5s %5s %6s %6s %8s %4s' % ('kernel', 'VGPR', 'AGPR', 'vspill', 'sspill', 'scratch', 'occ'))
for r in rows:
    name = r['name'].replace('cae::', '').split('(')[0][:70]
    print('%-70s %5s %5s %6s %6s %8s %4s' % (name, r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill', r.get('VGPR Spill')),
          r.get('SGPRs Spill', r.get('SGPR Spill')), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]')))
