"""The reference's own call pattern: many threads calling ``codec.encode(chunk)`` / ``codec.decode(buf)`` on one shared
codec (dask's threaded scheduler, ``src/compress.py:121-128``; bodies ``_autoencoders.py:539-584``).  The front door
coalesces the GPU part of concurrent calls; results must not depend on how calls were grouped."""
import struct
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest


# ---- host logic (no GPU): grouping, limits, error routing of the dispatcher ----------------------------------------
class _FakeDoor:
    """FrontDoor with the two GPU steps replaced: `_launch` records the batch, `_complete` answers (batch size, index).
    (The service loops pin the door's device first: the tests below stub torch.cuda.set_device.)"""

    def __new__(cls, max_batch=4, inflight=1, gate=None, fail_key=None):
        from cnn_autoencoder_amd import frontdoor as fd

        class Door(fd.FrontDoor):
            def __init__(self):  # no codec, no device
                import queue
                self.max_batch, self.inflight = max_batch, inflight
                self.dev = None
                self._q, self._cq = queue.Queue(), queue.Queue()
                self._slots = threading.local()
                self._sem = threading.Semaphore(inflight)
                self._started = self._closed = False
                self._start_mu = threading.Lock()
                self.batches = self.chunks = 0
                self._timers = []
                self.seen = []

            def _launch(self, batch):
                if gate is not None:
                    gate.wait()
                self.seen.append([r.key for r in batch])
                if fail_key is not None and batch[0].key[1] == fail_key:
                    raise ValueError('bad batch')
                return (batch,)

            def _complete(self, batch):
                for i, r in enumerate(batch):
                    r.result = (len(batch), i)
                    r.batch = None
                    r.done.set()

        return Door()


def test_dispatcher_groups_by_key_and_respects_max_batch(monkeypatch):
    import torch
    from cnn_autoencoder_amd import frontdoor as fd
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    gate = threading.Event()
    door = _FakeDoor(max_batch=4, inflight=1, gate=gate)
    # 6 requests of shape A, 3 of shape B queued while the dispatcher is held at its first launch
    with ThreadPoolExecutor(12) as pool:
        first = pool.submit(door._submit, fd._ENCODE, ('A',), {})
        while not door._started or door._q.qsize() > 0:  # the dispatcher took the first request and waits at the gate
            threading.Event().wait(0.01)
        futs = []
        for key in ['A'] * 6 + ['B'] * 3:
            futs.append(pool.submit(door._submit, fd._ENCODE, (key,), {}))
            while door._q.qsize() < len(futs):
                threading.Event().wait(0.005)
        gate.set()
        res = [f.result(timeout=20) for f in [first] + futs]
    door.close()
    sizes = [len(b) for b in door.seen]
    assert sizes == [1, 4, 2, 3], sizes  # first alone; A's split at max_batch; B's never mixed with A's
    assert all(len({k for k in b}) == 1 for b in door.seen)
    assert sorted(r.result[1] for r in res[1:5]) == [0, 1, 2, 3]


def test_dispatcher_routes_errors_to_the_callers_of_that_batch_only(monkeypatch):
    import torch
    from cnn_autoencoder_amd import frontdoor as fd
    monkeypatch.setattr(torch.cuda, 'set_device', lambda d: None)
    door = _FakeDoor(max_batch=8, inflight=2, fail_key='bad')
    with ThreadPoolExecutor(8) as pool:
        good = [pool.submit(door._submit, fd._ENCODE, ('ok',), {}) for _ in range(5)]
        bad = [pool.submit(door._submit, fd._DECODE, ('bad',), {}) for _ in range(2)]
        assert all(f.result(timeout=20).result is not None for f in good)
        for f in bad:
            with pytest.raises(ValueError, match='bad batch'):
                f.result(timeout=20)
        # the door keeps serving after a failed batch
        assert pool.submit(door._submit, fd._ENCODE, ('ok',), {}).result(timeout=20).result is not None
    door.close()
    with pytest.raises(RuntimeError):
        door._closed = True
        door._started = False
        door._ensure_started()


# ---- on the GPU --------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cae():
    import torch
    if not torch.cuda.is_available():
        pytest.skip('needs a HIP device')
    import cnn_autoencoder_amd as cae
    return cae


def _run_threads(n_threads, fn, items):
    with ThreadPoolExecutor(n_threads) as pool:
        return list(pool.map(fn, items))


@pytest.mark.gpu
def test_sixteen_threads_match_the_batched_side_door(cae):
    """16 threads x distinct tiles of two shapes, results matched by tile: encode() == encode_batch() byte for byte,
    decode() == decode_batch() pixel for pixel, whatever batches the door formed."""
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=64, channels_bn=96), seed=3)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    a = np.concatenate([synth.histo_tiles(24, 128, first_index=50), synth.uniform_tiles(8, 128)])
    b = synth.histo_tiles(16, 96, first_index=90)[:, :, :80].copy()  # 96 x 80
    ref = codec.encode_batch(a) + codec.encode_batch(b)
    tiles = list(a) + list(b)
    order = np.random.default_rng(0).permutation(len(tiles))
    got = _run_threads(16, lambda i: (i, codec.encode(tiles[i])), order)
    door = codec._front_door()
    assert door is not None and door.chunks == len(tiles)
    assert door.batches < len(tiles), 'no call was ever coalesced'
    for i, buf in got:
        assert buf == ref[i], f'tile {i}: payload differs from encode_batch'
    rec_ref = list(codec.decode_batch(ref[:len(a)])) + list(codec.decode_batch(ref[len(a):]))

    def dec(i):
        if i % 2:
            return i, codec.decode(ref[i])
        out = np.empty_like(rec_ref[i])
        assert codec.decode(ref[i], out=out) is out
        return i, out
    for i, rec in _run_threads(16, dec, order):
        assert rec.dtype == np.uint8 and np.array_equal(rec, rec_ref[i]), f'tile {i}: reconstruction differs'
    # both directions at once on the one instance
    mixed = _run_threads(16, lambda i: codec.encode(tiles[i]) if i % 3 else codec.decode(ref[i]), order)
    for i, r in zip(order, mixed):
        assert (r == ref[i]) if i % 3 else np.array_equal(r, rec_ref[i])
    codec.close()


@pytest.mark.gpu
def test_full_size_tiles_through_the_door(cae):
    """BASELINE's tile size, canonical model, the default arithmetic: 8 threads, 12 tiles of 1024 x 1024 x 3."""
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(synth.CANONICAL, seed=0)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    tiles = synth.histo_tiles(12, 1024, first_index=7)
    ref = codec.encode_batch(tiles)
    got = _run_threads(8, codec.encode, list(tiles))
    assert got == ref
    rec_ref = codec.decode_batch(ref)
    rec = _run_threads(8, codec.decode, ref)
    assert all(np.array_equal(r, rr) for r, rr in zip(rec, rec_ref))
    codec.close()


@pytest.mark.gpu
def test_errors_reach_only_their_caller(cae):
    from cnn_autoencoder_amd import synth, _lib
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=2)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    tiles = synth.uniform_tiles(8, 64)
    ref = codec.encode_batch(tiles)
    corrupt = ref[0][:16] + ref[0][16:40]  # header of a 64 x 64 tile, payload cut short

    def job(i):
        if i == 3:
            with pytest.raises(ValueError):
                codec.encode(np.zeros((64, 64, 4), np.uint8))  # wrong channel count
            return None
        if i == 5:
            with pytest.raises((_lib.CaeError, ValueError)):
                codec.decode(corrupt)
            return None
        return codec.encode(tiles[i])
    got = _run_threads(8, job, range(8))
    assert all(g == ref[i] for i, g in enumerate(got) if g is not None)
    codec.close()


@pytest.mark.gpu
def test_range_guard_repeat_through_the_door(cae):
    """A batch whose activations leave the f16 range is repeated on the fp32 kernels by the completer; callers see
    the same bytes as the batched side door (tests/test_range_guard.py holds the guard itself)."""
    import torch
    from cnn_autoencoder_amd import synth
    # (the model of test_range_guard.py::test_pipelined_drivers_repeat_only_the_batches_that_overflow: bright tiles
    # leave the f16 range in an activation-free track)
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3, act_layer_type=None)
    state = synth.synthetic_state(cfg, seed=15)
    state['encoder']['analysis_track.0.model.0.weight'] *= 5.0e4
    state['encoder']['analysis_track.2.model.0.weight'] *= 2.0e-5
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    tiles = np.random.default_rng(5).integers(200, 256, (6, 64, 96, 3), dtype=np.uint8)
    ref = codec.encode_batch(tiles)
    enc = codec._model['encoder'].module
    before = enc.fp32_fallbacks
    got = _run_threads(6, codec.encode, list(tiles))
    assert got == ref
    if enc.precision_code() == 1:
        assert enc.fp32_fallbacks > before, 'the stress weights did not trip the guard'
    assert torch.cuda.is_available()
    codec.close()
