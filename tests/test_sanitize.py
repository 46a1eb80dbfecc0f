"""The host range coder (the product's cae_entropy.cpp, unchanged) under AddressSanitizer + UBSan on the CPU: random
tables, bypass symbols, every lockstep width, damaged and truncated streams.  GPU sanitizers are not available on the
pool; this is the part of the path where a memory error would corrupt somebody's zarr store silently."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_coder_under_asan_and_ubsan(tmp_path):
    cxx = shutil.which('g++')
    if cxx is None:
        pytest.skip('g++ not available')
    exe = str(tmp_path / 'entropy_sanitize')
    cmd = [cxx, '-std=c++17', '-O1', '-g', '-fsanitize=address,undefined', '-fno-sanitize-recover=undefined', '-pthread',
           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'cnn_autoencoder_amd', 'csrc'),
           os.path.join(ROOT, 'tests', 'native', 'entropy_sanitize.cpp'),
           os.path.join(ROOT, 'cnn_autoencoder_amd', 'csrc', 'cae_entropy.cpp'), '-o', exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    for width in ('1', '2', '4'):
        env = dict(os.environ, CAE_CODER_LOCKSTEP=width, ASAN_OPTIONS='detect_leaks=1:abort_on_error=0')
        r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert f'ok (lockstep {width})' in r.stdout
