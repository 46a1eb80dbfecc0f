"""Rate of the reference's own call pattern: T Python threads calling ``codec.encode(chunk)`` / ``codec.decode(buf)``
one chunk at a time on one shared codec, as dask's threaded scheduler does (``src/compress.py:121-128``,
``src/decompress.py:51-58``).  Used by ``bench.py`` (the ``dropin`` object of the bench line) and ``tools/bench_dropin.py``.
"""
from __future__ import annotations

import time
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Sequence

import numpy as np


def _drive(fn, items, threads: int, repeat: int):
    """calls fn(item) for every item `repeat` times from `threads` threads -> (seconds, process CPU seconds, results
    of the last pass)"""
    seq = list(range(len(items))) * repeat
    with ThreadPoolExecutor(threads) as pool:
        list(pool.map(fn, items[:threads]))  # threads started, pinned slots allocated
        c0, t0 = time.process_time(), time.perf_counter()
        res = list(pool.map(lambda i: fn(items[i]), seq))
        dt, cpu = time.perf_counter() - t0, time.process_time() - c0
    return dt, cpu, res[-len(items):]


def measure(codec, tiles: np.ndarray, thread_counts: Sequence[int] = (1, 8, 16), budget_s: float = 2.0) -> Dict:
    """tiles (n,h,w,c) uint8 in HOST memory (the codec contract hands over host chunks: PCIe is inside the number).
    -> {'encode': {T: tiles/s}, 'decode': {T: tiles/s}, ...}; payloads of every pass are checked against the batched
    side door, byte for byte."""
    tiles = np.ascontiguousarray(tiles)
    items = list(tiles)
    ref = codec.encode_batch(tiles)
    rec_ref = codec.decode_batch(ref)
    out = dict(tile=list(tiles.shape[1:]), distinct_tiles=len(items), encode={}, decode={}, cpus_busy={}, identical=True)
    door = codec._front_door()
    for T in thread_counts:
        for name, fn, data in (('encode', codec.encode, items), ('decode', codec.decode, ref)):
            dt1, _, _ = _drive(fn, data, T, 1)  # also the warm-up of this thread count
            repeat = int(max(1, min(64, budget_s / max(dt1, 1e-3))))
            if door is not None:
                door.stats(reset=True)
            dt, cpu, res = _drive(fn, data, T, repeat)
            out[name][str(T)] = repeat * len(data) / dt
            out['cpus_busy'][f'{name}{T}'] = cpu / dt
            if door is not None:
                st = door.stats()
                out.setdefault('mean_batch', {})[f'{name}{T}'] = st['chunks'] / max(1.0, st['batches'])
                # a call's milliseconds inside the library, averaged over the calls: staging (copy into pinned memory /
                # range decode), waiting (split: queue, launch, device, pull, wake-up), coding (range encode / copy out)
                out.setdefault('ms_per_call', {})[f'{name}{T}'] = {
                    k: round(1e3 * v / max(1.0, st['chunks']), 4) for k, v in st.items()
                    if k not in ('batches', 'chunks', 'fp32_repeats')}
                out.setdefault('wall_ms_per_chunk', {})[f'{name}{T}'] = round(1e3 * dt / (repeat * len(data)), 4)
            same = (res == ref) if name == 'encode' else all(np.array_equal(a, b) for a, b in zip(res, rec_ref))
            out['identical'] = bool(out['identical'] and same)
    return out
