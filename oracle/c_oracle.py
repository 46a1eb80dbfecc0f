"""ctypes binding of oracle/_build/liboracle.so (plain-C restatement, rans64_oracle.c).

TEST INFRASTRUCTURE ONLY -- see rans64_oracle.c header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(['make', '-s', '-C', _HERE, 'all'])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, '_build', 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = ctypes.CDLL(path)
        L.oracle_pmf_to_quantized_cdf.restype = ctypes.c_int
        L.oracle_rans_encode_with_indexes.restype = ctypes.c_long
        L.oracle_rans_decode_with_indexes.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(pmf, dtype=np.float32)
    cdf = np.zeros(len(pmf) + 1, dtype=np.uint32)
    rc = lib().oracle_pmf_to_quantized_cdf(_p(pmf, ctypes.c_float), ctypes.c_int(len(pmf)),
                                           ctypes.c_int(precision), _p(cdf, ctypes.c_uint32))
    if rc != 0:
        raise ValueError('oracle_pmf_to_quantized_cdf failed: %d' % rc)
    return cdf.astype(np.int64).tolist()


def _tables(cdfs, cdf_lengths, offsets):
    cdfs = np.ascontiguousarray(cdfs, dtype=np.int32)
    return cdfs, np.ascontiguousarray(cdf_lengths, dtype=np.int32), np.ascontiguousarray(offsets, dtype=np.int32)


def rans_encode_with_indexes(symbols, indexes, cdfs, cdf_lengths, offsets) -> bytes:
    symbols = np.ascontiguousarray(symbols, dtype=np.int32)
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    cdfs, lens, offs = _tables(cdfs, cdf_lengths, offsets)
    cap = 4 * (4 * len(symbols) + 64)
    out = np.empty(cap, dtype=np.uint8)
    n = lib().oracle_rans_encode_with_indexes(
        _p(symbols, ctypes.c_int32), _p(indexes, ctypes.c_int32), ctypes.c_long(len(symbols)),
        _p(cdfs, ctypes.c_int32), ctypes.c_int(cdfs.shape[1]), _p(lens, ctypes.c_int32),
        _p(offs, ctypes.c_int32), _p(out, ctypes.c_uint8), ctypes.c_long(cap))
    if n == -2:
        raise ValueError('symbol outside the codable range')
    if n < 0:
        raise RuntimeError('oracle encode failed')
    return out[:n].tobytes()


def rans_decode_with_indexes(encoded: bytes, indexes, cdfs, cdf_lengths, offsets):
    indexes = np.ascontiguousarray(indexes, dtype=np.int32)
    cdfs, lens, offs = _tables(cdfs, cdf_lengths, offsets)
    buf = np.frombuffer(bytes(encoded) + b'\0' * 8, dtype=np.uint8).copy()
    out = np.empty(len(indexes), dtype=np.int32)
    rc = lib().oracle_rans_decode_with_indexes(
        _p(buf, ctypes.c_uint8), ctypes.c_long(len(encoded)), _p(indexes, ctypes.c_int32),
        ctypes.c_long(len(indexes)), _p(cdfs, ctypes.c_int32), ctypes.c_int(cdfs.shape[1]),
        _p(lens, ctypes.c_int32), _p(offs, ctypes.c_int32), _p(out, ctypes.c_int32))
    if rc != 0:
        raise RuntimeError('oracle decode failed')
    return out.tolist()
