"""Micro-batching front door of the numcodecs ``encode`` / ``decode`` contract.

The reference's callers never see a batch: dask's threaded scheduler runs one task per zarr chunk and every task
calls ``codec.encode(chunk)`` / ``codec.decode(buf)`` on ONE shared codec instance (``src/compress.py:121-128``,
``src/decompress.py:51-58``; ``src/models/tasks/_autoencoders.py:539-555`` and ``:557-584`` are the bodies, batch 1,
``:544``).  This module keeps that blocking one-chunk contract and makes it fast:

* the range coder of a chunk runs in the CALLER's thread (libcae_hip.so releases the GIL), outside every lock, so a
  pool of N dask threads is N coder cores;
* the GPU part of concurrent calls is coalesced: a caller parks its tile (pinned, thread-local) in a queue, one
  dispatcher thread launches the analysis / synthesis for whatever is queued (up to ``max_batch`` chunks of one
  shape) while at most ``inflight`` earlier batches are still on the device -- requests pile up exactly as long as
  the GPU is busy, so nothing waits on a timer and a lone caller is served at once (batch of one);
* a completer thread waits for the batch (blocking HIP event), pulls the result over the DMA engines into one
  pinned buffer and wakes the callers, each of which takes its own slice.

Results do not depend on how calls were grouped: every kernel of the path works on a tile by itself
(tests/test_frontdoor.py compares 16 threads with ``encode_batch`` byte for byte).
"""
from __future__ import annotations

import os
import queue
import struct
import threading
import time
import weakref
from typing import List, Optional

import numpy as np
import torch

from . import _lib

_ENCODE, _DECODE = 0, 1


class _Request:
    __slots__ = ('kind', 'key', 'slot', 'done', 'error', 'result', 'batch', 'stamps')

    def __init__(self, kind, key, slot):
        self.kind, self.key, self.slot = kind, (kind,) + tuple(key), slot
        self.done = threading.Event()
        self.error: Optional[BaseException] = None
        self.result = None
        self.batch = None
        self.stamps = [time.perf_counter(), 0.0, 0.0, 0.0, 0.0]  # submitted, picked, launched, ready, pulled


class _PinnedBatch:
    """One pinned result buffer shared by the callers of a batch; goes back to the pool when the last one let go."""

    def __init__(self, pool, key, tensor, users: int):
        self._pool, self._key, self.tensor = pool, key, tensor
        self.array = tensor.numpy()
        self._users = users
        self._mu = threading.Lock()

    def release(self):
        with self._mu:
            self._users -= 1
            last = self._users == 0
        if last:
            self._pool.put(self._key, self.tensor)


class _PinnedPool:
    """Free lists of pinned host tensors by (shape, dtype) (hipHostMalloc of a batch costs milliseconds)."""

    def __init__(self, keep: int = 4):
        self._free = {}
        self._mu = threading.Lock()
        self._keep = keep

    def get(self, shape, dtype, users: int) -> _PinnedBatch:
        key = (tuple(shape), dtype)
        with self._mu:
            lst = self._free.get(key)
            t = lst.pop() if lst else None
        if t is None:
            t = torch.empty(shape, dtype=dtype, pin_memory=True)
        return _PinnedBatch(self, key, t, users)

    def put(self, key, tensor):
        with self._mu:
            lst = self._free.setdefault(key, [])
            if len(lst) < self._keep:
                lst.append(tensor)


class FrontDoor:
    """Blocking ``encode(tile) -> bytes`` / ``decode(buf) -> ndarray`` for many threads on one codec."""

    def __init__(self, codec, max_batch: int = 0, inflight: int = 0):
        from .codec import _module
        self._codec_ref = weakref.ref(codec)
        self._lock = codec._lock  # the GPU section: the tracks' workspaces are used in stream order
        self.enc = _module(codec._model['encoder'])
        self.dec = _module(codec._model['decoder'])
        self.eb = _module(codec._model['fact_ent'])
        self.level = len(self.dec.synthesis_track)
        self.max_batch = max_batch or int(os.environ.get('CAE_DOOR_BATCH', '32'))
        self.inflight = inflight or int(os.environ.get('CAE_DOOR_INFLIGHT', '2'))
        self.dev = _lib.require_gpu()
        self._q: 'queue.Queue' = queue.Queue()
        self._cq: 'queue.Queue' = queue.Queue()
        self._slots = threading.local()
        self._pool = _PinnedPool()
        self._sem = threading.Semaphore(self.inflight)
        self._up = None
        self._started = False
        self._closed = False
        self._start_mu = threading.Lock()
        self.batches = 0  # (statistics: batches launched / chunks served)
        self.chunks = 0
        self._timers: List[dict] = []  # one dict of accumulated seconds per thread that used the door (timers())
        with self._lock:
            self.eb._sync_handle()  # tables on the library side before many threads read them

    # ---- where the time goes ------------------------------------------------------------------------------------
    def _tm(self) -> dict:
        d = getattr(self._slots, 'tm', None)
        if d is None:
            d = self._slots.tm = {}
            with self._start_mu:
                self._timers.append(d)
        return d

    def timers(self, reset: bool = False) -> dict:
        """Seconds summed over all threads: callers' `stage` (tile into pinned memory / range decode), `wait` (queue +
        GPU + pull), `code` (range encode / copy out); dispatcher `slot` (waiting for an in-flight slot), `launch`;
        completer `sync` (waiting for the batch's kernels), `pull` (DMA to the host)."""
        out: dict = {}
        with self._start_mu:
            for d in self._timers:
                for k, v in list(d.items()):
                    out[k] = out.get(k, 0.0) + v
                if reset:
                    d.clear()
        return out

    # ---- threads -----------------------------------------------------------------------------------------------
    def _ensure_started(self):
        if self._started:
            return
        with self._start_mu:
            if self._started:
                return
            if self._closed:
                raise RuntimeError('front door is closed')
            self._threads = [threading.Thread(target=self._dispatch_loop, name='cae-door-dispatch', daemon=True),
                             threading.Thread(target=self._complete_loop, name='cae-door-complete', daemon=True)]
            for t in self._threads:
                t.start()
            self._started = True

    def close(self):
        """Stops the two service threads (pending calls are served first)."""
        with self._start_mu:
            if self._closed:
                return
            self._closed = True
            started = self._started
        if started:
            self._q.put(None)
            for t in self._threads:
                t.join(timeout=30)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _fail(batch: List[_Request], err: BaseException):
        for r in batch:
            r.error = err
            r.done.set()

    def _dispatch_loop(self):
        try:
            self._dispatch()
        finally:  # (also after an unexpected error: nobody may be left waiting)
            self._closed = True
            self._cq.put(None)
            while True:
                try:
                    r = self._q.get_nowait()
                except queue.Empty:
                    break
                if r is not None:
                    self._fail([r], RuntimeError('front door closed'))

    def _dispatch(self):
        torch.cuda.set_device(self.dev)
        held = None
        stop = False
        while not stop:
            first = held if held is not None else self._q.get()
            held = None
            if first is None:
                break
            tm = self._tm()
            t0 = time.perf_counter()
            self._sem.acquire()  # requests keep piling up while `inflight` batches are on the device
            t1 = time.perf_counter()
            tm['slot'] = tm.get('slot', 0.0) + t1 - t0
            batch = [first]
            while len(batch) < self.max_batch:
                try:
                    nxt = self._q.get_nowait()
                except queue.Empty:
                    break
                if nxt is None:
                    stop = True
                    break
                if nxt.key != first.key:
                    held = nxt  # another shape / direction: next batch
                    break
                batch.append(nxt)
            t1 = time.perf_counter()
            for r in batch:
                r.stamps[1] = t1
            try:
                item = self._launch(batch)
            except BaseException as e:  # noqa: BLE001 - handed to the callers
                self._sem.release()
                self._fail(batch, e)
                continue
            t2 = time.perf_counter()
            for r in batch:
                r.stamps[2] = t2
            tm['launch'] = tm.get('launch', 0.0) + t2 - t1
            self._cq.put(item)
        if held is not None:
            self._fail([held], RuntimeError('front door closed'))

    def _complete_loop(self):
        torch.cuda.set_device(self.dev)
        while True:
            item = self._cq.get()
            if item is None:
                break
            batch = item[0]
            try:
                self._complete(*item)
            except BaseException as e:  # noqa: BLE001
                self._fail(batch, e)
            finally:
                self._sem.release()

    # ---- GPU section -------------------------------------------------------------------------------------------
    def _up_stream(self):
        if self._up is None:
            self._up = torch.cuda.Stream(self.dev)
        return self._up

    @torch.no_grad()
    def _launch(self, batch: List[_Request]):
        kind = batch[0].kind
        n = len(batch)
        up = self._up_stream()
        with self._lock:
            main = torch.cuda.current_stream(self.dev)
            # the callers' pinned buffers go up on the side stream, beside the kernels of the batch before
            with torch.cuda.stream(up):
                first = batch[0].slot[kind]
                x = torch.empty((n,) + tuple(first.shape), dtype=first.dtype, device=self.dev)
                for i, r in enumerate(batch):
                    x[i].copy_(r.slot[kind], non_blocking=True)
                landed = torch.cuda.Event()
                landed.record(up)
            main.wait_event(landed)
            x.record_stream(main)
            if kind == _ENCODE:
                out, guard = self.enc.forward_u8_symbols(x, self.eb, defer=True)
            else:
                out, guard = self.dec.forward_symbols_u8(x, self.eb, defer=True)
            ready = torch.cuda.Event(blocking=True)
            ready.record(main)
        self.batches += 1
        self.chunks += n
        return batch, out, guard, ready, x

    def _complete(self, batch, out, guard, ready, x):
        tm = self._tm()
        t0 = time.perf_counter()
        ready.synchronize()
        t1 = t0r = time.perf_counter()
        tm['sync'] = tm.get('sync', 0.0) + t1 - t0
        if guard.overflowed():  # f16x3 range guard (rare): this batch again on the exact-fp32 kernels
            with self._lock, torch.no_grad():
                out = guard.rerun()
                torch.cuda.current_stream(self.dev).synchronize()
        n = len(batch)
        per = tuple(out.shape[1:])
        pinned = self._pool.get((n,) + per, out.dtype, users=n)
        t1 = time.perf_counter()
        _lib.check(_lib.lib().cae_copy_to_host(pinned.tensor.data_ptr(), out.data_ptr(), out.numel() * out.element_size()))
        t2 = time.perf_counter()
        tm['pull'] = tm.get('pull', 0.0) + t2 - t1
        for r in batch:
            r.stamps[3], r.stamps[4] = t0r, t2
        for i, r in enumerate(batch):
            r.result = pinned.array[i]
            r.batch = pinned
            r.done.set()

    # ---- caller side -------------------------------------------------------------------------------------------
    def _slot(self, kind, shape, dtype) -> dict:
        slot = getattr(self._slots, 'slot', None)
        if slot is None:
            slot = self._slots.slot = {}
        t = slot.get(kind)
        if t is None or tuple(t.shape) != tuple(shape):
            t = slot[kind] = torch.empty(tuple(shape), dtype=dtype, pin_memory=True)
        return slot

    def _submit(self, kind, key, slot) -> _Request:
        self._ensure_started()
        req = _Request(kind, key, slot)
        self._q.put(req)
        while not req.done.wait(1.0):
            if not self._threads[0].is_alive() and not req.done.is_set():
                raise RuntimeError('front door closed')
        if req.error is not None:
            raise req.error
        tm = self._tm()  # a call's way through the door: queue, launch, device, pull, wake-up
        st = req.stamps + [time.perf_counter()]
        for k, name in enumerate(('w_queue', 'w_launch', 'w_device', 'w_pull', 'w_wake')):
            tm[name] = tm.get(name, 0.0) + st[k + 1] - st[k]
        return req

    def encode(self, tile: np.ndarray) -> bytes:
        """(h,w,c) uint8 chunk -> ``>QQ`` header + rANS payload (``_autoencoders.py:539-555``)."""
        tile = np.asarray(tile)
        if tile.ndim != 3 or tile.dtype != np.uint8:
            raise ValueError(f'expected a uint8 (h,w,c) chunk, got {tile.dtype} {tile.shape}')
        h, w, c = tile.shape
        if c != self.enc._dims[0]:
            raise ValueError(f'expected uint8 (h,w,{self.enc._dims[0]}), got {tile.shape}')
        tm = self._tm()
        t0 = time.perf_counter()
        slot = self._slot(_ENCODE, tile.shape, torch.uint8)
        np.copyto(slot[_ENCODE].numpy(), tile)  # into pinned memory, in this thread
        t1 = time.perf_counter()
        req = self._submit(_ENCODE, tile.shape, slot)
        t2 = time.perf_counter()
        try:
            sym = req.result  # (C, lh, lw) int32 in the batch's pinned buffer
            payload = self.eb.encode_symbols(sym.reshape(1, sym.shape[0], -1), threads=1)[0]
        finally:
            req.batch.release()
        res = struct.pack('>QQ', h, w) + payload
        t3 = time.perf_counter()
        for k, v in (('stage', t1 - t0), ('wait', t2 - t1), ('code', t3 - t2)):
            tm[k] = tm.get(k, 0.0) + v
        return res

    def decode(self, buf, out=None):
        """chunk bytes -> (h,w,c) uint8 tile (``_autoencoders.py:557-584``); ``out``: written in place when given."""
        tm = self._tm()
        t0 = time.perf_counter()
        view = memoryview(buf)
        if view.nbytes < 16:
            raise ValueError('chunk shorter than its 16-byte header')
        h, w = struct.unpack('>QQ', view[:16])
        lh, lw = h // 2 ** self.level, w // 2 ** self.level
        C = self.eb.channels
        slot = self._slot(_DECODE, (C, lh, lw), torch.int32)
        sym = slot[_DECODE].numpy().reshape(1, C, lh * lw)
        self.eb.decode_symbols([bytes(view[16:])], lh * lw, threads=1, out=sym)  # range decoder in this thread
        t1 = time.perf_counter()
        req = self._submit(_DECODE, (C, lh, lw), slot)
        t2 = time.perf_counter()
        try:
            rec = req.result  # (H,W,c) uint8 in the batch's pinned buffer
            if out is None:
                res = rec.copy()
            else:
                res = np.asarray(out)
                np.copyto(res.reshape(-1).view(np.uint8), rec.reshape(-1))
        finally:
            req.batch.release()
        t3 = time.perf_counter()
        for k, v in (('stage', t1 - t0), ('wait', t2 - t1), ('code', t3 - t2)):
            tm[k] = tm.get(k, 0.0) + v
        return res
