import sys, time, ctypes, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cnn_autoencoder_amd import _lib
L = _lib.lib()
d = np.load('gpurun_out/syms.npz')
sym4 = d['sym']  # (4,192,64,64)
# 32 distinct streams: flips/rolls of the 4
streams = []
for k in range(32):
    s = sym4[k % 4]
    if k & 4: s = s[:, ::-1]
    if k & 8: s = s[:, :, ::-1]
    if k & 16: s = s.transpose(0, 2, 1)
    streams.append(np.ascontiguousarray(s))
sym = np.ascontiguousarray(np.stack(streams).reshape(32, 192, 4096)).astype(np.int32)
h = _lib.Handle(3, 128, 192, 4, 3)
cdf = np.ascontiguousarray(d['cdf'], dtype=np.int32); ln = np.ascontiguousarray(d['length'], dtype=np.int32)
off = np.ascontiguousarray(d['offset'], dtype=np.int32); med = np.ascontiguousarray(d['medians'], dtype=np.float32)
_lib.check(L.cae_model_set_entropy(h.ptr, 192, cdf.shape[1], cdf.ctypes.data, ln.ctypes.data, off.ctypes.data, med.ctypes.data))
threads = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = 32
bufs = (ctypes.c_void_p * n)(); lens = (ctypes.c_size_t * n)()
def enc():
    _lib.check(L.cae_rans_encode_batch(h.ptr, sym.ctypes.data, n, 4096, bufs, lens, threads))
enc()
pay = [ctypes.string_at(bufs[i], lens[i]) for i in range(n)]
import hashlib
print('digest', hashlib.sha1(b''.join(pay)).hexdigest(), 'bytes', sum(lens), 'bits/sym', 8*sum(lens)/sym.size)
for i in range(n): L.cae_free(bufs[i])
reps = 5
t0 = time.perf_counter()
for _ in range(reps):
    enc()
    for i in range(n): L.cae_free(bufs[i])
te = (time.perf_counter() - t0) / reps
arr = (ctypes.c_char_p * n)(*pay); ls = (ctypes.c_size_t * n)(*[len(p) for p in pay])
out = np.empty_like(sym)
def dec():
    _lib.check(L.cae_rans_decode_batch(h.ptr, arr, ls, n, 4096, out.ctypes.data, threads))
dec(); assert np.array_equal(out, sym)
t0 = time.perf_counter()
for _ in range(reps): dec()
td = (time.perf_counter() - t0) / reps
print(f'threads {threads}: encode {1e3*te:.1f} ms ({1e9*te*threads/sym.size:.2f} ns/sym/thread)  decode {1e3*td:.1f} ms ({1e9*td*threads/sym.size:.2f} ns/sym/thread)')
