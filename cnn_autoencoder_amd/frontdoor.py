"""Front door of the numcodecs ``encode`` / ``decode`` contract: binding of ``cae_door_*`` (include/cae_hip.h).

The reference's callers never see a batch: dask's threaded scheduler runs one task per zarr chunk and every task
calls ``codec.encode(chunk)`` / ``codec.decode(buf)`` on ONE shared codec instance (``src/compress.py:121-128``,
``src/decompress.py:51-58``; ``src/models/tasks/_autoencoders.py:539-555`` and ``:557-584`` are the bodies, batch 1,
``:544``).  The library keeps that blocking one-chunk contract and makes it fast (csrc/cae_door.hip):

* a call is ONE ctypes call (the GIL is released for all of it): the chunk is staged in pinned memory and range-coded
  in the caller's thread, outside every lock, so a pool of N dask threads is N coder cores;
* the GPU part of the calls that are waiting at the same time runs as one batch: a dispatcher thread launches the
  analysis / synthesis for whatever is queued (chunks of one shape, at most ``max_batch``) as soon as one of
  ``inflight`` device-side slots is free -- requests pile up exactly while the GPU is busy, nothing waits on a timer,
  a lone caller is served at once; a completer thread pulls the batch over the DMA engines and wakes the callers.

Results do not depend on how calls were grouped: every kernel of the path works on a tile by itself
(tests/test_frontdoor.py compares 16 threads with ``encode_batch`` byte for byte).
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict

import numpy as np

from . import _lib

_STAT_NAMES = ('batches', 'chunks', 'fp32_repeats', 'stage', 'wait', 'code', 'w_queue', 'w_launch', 'w_device', 'w_pull',
               'w_wake', 'queued')


class FrontDoor:
    """Blocking ``encode(tile) -> bytes`` / ``decode(buf) -> ndarray`` for many threads on one codec."""

    def __init__(self, codec, max_batch: int = 0, inflight: int = 0):
        from .codec import _module
        self._lock = codec._lock  # uploads of changed parameters (the calls themselves need no lock)
        self.enc = _module(codec._model['encoder'])
        self.dec = _module(codec._model['decoder'])
        self.eb = _module(codec._model['fact_ent'])
        self.max_batch = max_batch or int(os.environ.get('CAE_DOOR_BATCH', '32'))
        self.inflight = inflight or int(os.environ.get('CAE_DOOR_INFLIGHT', '3'))
        _lib.require_gpu()
        self._ver = None
        self._door = _lib.c_void_p()
        self._sync()
        _lib.check(_lib.lib().cae_door_create(self.enc._handle.ptr, self.dec._handle.ptr, self.max_batch, self.inflight,
                                              ctypes.byref(self._door)))

    def _sync(self):
        """Parameters / tables that changed since the last call go to the two track handles first (the modules' own
        upload path); a plain version compare otherwise."""
        ver = (self.enc._param_versions(), self.dec._param_versions(), self.eb.tables_version())
        if ver != self._ver:
            with self._lock:
                self.enc._sync_entropy(self.eb)
                self.dec._sync_entropy(self.eb)
                self._ver = ver

    def close(self):
        """Serves the queued calls, stops the two service threads; no call may be running."""
        door, self._door = self._door, _lib.c_void_p()
        if door:
            _lib.lib().cae_door_destroy(door)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def stats(self, reset: bool = False) -> Dict[str, float]:
        """batches launched, chunks served, batches repeated on fp32; seconds summed over the calls: callers' `stage`
        (tile into pinned memory / range decode), `wait`, `code` (range encode / copy out), and `wait` split into
        `w_queue`, `w_launch`, `w_device`, `w_pull` (DMA to the host), `w_wake`."""
        buf = (ctypes.c_double * len(_STAT_NAMES))()
        _lib.check(_lib.lib().cae_door_stats(self._door, buf, len(_STAT_NAMES), int(reset)))
        return dict(zip(_STAT_NAMES, buf))

    def hold(self, on: bool = True):
        """While held the dispatcher starts no batch: calls queue up (``stats()['queued']``) and go out together --
        ``max_batch`` at a time -- on ``hold(False)``."""
        _lib.check(_lib.lib().cae_door_hold(self._door, int(bool(on))))

    @property
    def batches(self) -> int:
        return int(self.stats()['batches'])

    @property
    def chunks(self) -> int:
        return int(self.stats()['chunks'])

    def encode(self, tile: np.ndarray) -> bytes:
        """(h,w,c) uint8 chunk -> ``>QQ`` header + rANS payload (``_autoencoders.py:539-555``)."""
        tile = np.asarray(tile)
        if tile.ndim != 3 or tile.dtype != np.uint8:
            raise ValueError(f'expected a uint8 (h,w,c) chunk, got {tile.dtype} {tile.shape}')
        if not tile.flags.c_contiguous:
            tile = np.ascontiguousarray(tile)
        self._sync()
        h, w, c = tile.shape
        out, n = _lib.c_void_p(), _lib.c_size_t()
        L = _lib.lib()
        _lib.check(L.cae_door_encode(self._door, tile.ctypes.data, h, w, c, ctypes.byref(out), ctypes.byref(n)))
        try:
            return ctypes.string_at(out.value, n.value)
        finally:
            L.cae_free(out)

    def decode(self, buf, out=None) -> np.ndarray:
        """chunk bytes -> (H,W,c) uint8 tile (``_autoencoders.py:557-584``); written into ``out`` when it is a
        contiguous array of exactly that many bytes (then ``out`` itself is returned)."""
        if isinstance(buf, bytes):
            ptr, n, keep = buf, len(buf), buf
        else:
            keep = np.frombuffer(buf, dtype=np.uint8)
            ptr, n = keep.ctypes.data, keep.nbytes
        self._sync()
        L = _lib.lib()
        h, w, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        _lib.check(L.cae_door_decode_shape(self._door, ptr, n, ctypes.byref(h), ctypes.byref(w), ctypes.byref(c)))
        nbytes = h.value * w.value * c.value
        if isinstance(out, np.ndarray) and out.nbytes == nbytes and out.flags.c_contiguous and out.flags.writeable:
            res = out
        else:
            res = np.empty((h.value, w.value, c.value), dtype=np.uint8)
        _lib.check(L.cae_door_decode(self._door, ptr, n, res.ctypes.data, nbytes))
        del keep
        return res
