#!/usr/bin/env python3
"""List the gfx950 kernels of libcae_hip.so that use scratch (private) memory or spill registers.

The conv / deconv kernels keep 64-256 accumulator registers per lane; twice in round 1 an innocent-looking change to
the store epilogue made the compiler keep them in a scratch-memory array instead (3x slower layer, no spill count
reported).  This scans the embedded code objects (ELF images inside .hip_fatbin) with llvm-readelf and prints
`kernel  private_segment_fixed_size  vgpr_spill_count` for every kernel whose values are non-zero.

usage: scratch_check.py [path/to/libcae_hip.so]      (exit status 0 always; tests/test_host.py asserts on hot kernels)"""
import os, re, struct, subprocess, sys, tempfile

READELF = '/opt/rocm/lib/llvm/bin/llvm-readelf'


def embedded_elfs(blob):
    pos = 0
    while True:
        i = blob.find(b'\x7fELF\x02\x01\x01', pos)
        if i < 0:
            return
        e_machine = struct.unpack_from('<H', blob, i + 18)[0]
        e_shoff, = struct.unpack_from('<Q', blob, i + 40)
        e_shentsize, e_shnum = struct.unpack_from('<HH', blob, i + 58)
        size = e_shoff + e_shentsize * e_shnum
        if e_machine == 224 and 0 < size <= len(blob) - i:  # EM_AMDGPU
            yield blob[i:i + size]
            pos = i + size
        else:
            pos = i + 4


def kernel_table(path):
    blob = open(path, 'rb').read()
    out = {}
    for img in embedded_elfs(blob):
        with tempfile.NamedTemporaryFile(suffix='.co') as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([READELF, '--notes', f.name], capture_output=True, text=True).stdout
        for block in txt.split('- .agpr_count:')[1:]:
            name = re.search(r'\.name:\s+(\S+)', block)
            priv = re.search(r'\.private_segment_fixed_size:\s+(\d+)', block)
            spill = re.search(r'\.vgpr_spill_count:\s+(\d+)', block)
            if name and priv:
                out[name.group(1)] = (int(priv.group(1)), int(spill.group(1)) if spill else 0)
    return out


def demangle(names):
    try:
        return subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.split('\n')
    except OSError:
        return list(names)


if __name__ == '__main__':
    so = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            'cnn_autoencoder_amd', 'libcae_hip.so')
    table = kernel_table(so)
    names = list(table)
    print(f'{len(table)} kernels in {so}')
    for mangled, nice in zip(names, demangle(names)):
        priv, spill = table[mangled]
        if priv or spill:
            print(f'{priv:6d} B scratch  {spill:4d} spilled VGPRs  {nice}')
