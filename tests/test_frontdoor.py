"""The reference's own call pattern: many threads calling ``codec.encode(chunk)`` / ``codec.decode(buf)`` on one shared
codec (dask's threaded scheduler, ``src/compress.py:121-128``; bodies ``_autoencoders.py:539-584``).  The front door
coalesces the GPU part of concurrent calls; results must not depend on how calls were grouped."""
import struct
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest


# ---- host side (no GPU): the C ABI of the door is exported and refuses what it must ------------------------------------
def test_door_symbols_and_argument_checks(built_lib):
    import ctypes
    from cnn_autoencoder_amd import _lib
    L = _lib.lib()
    for name in ('cae_door_create', 'cae_door_destroy', 'cae_door_encode', 'cae_door_decode_shape', 'cae_door_decode',
                 'cae_door_stats'):
        assert hasattr(L, name)
    door = ctypes.c_void_p()
    assert L.cae_door_create(None, None, 0, 0, ctypes.byref(door)) == -1  # CAE_ERR_ARG, no device touched
    assert b'NULL' in L.cae_last_error()
    assert L.cae_door_encode(None, None, 4, 4, 3, None, None) == -1
    assert L.cae_door_decode(None, None, 0, None, 0) == -1
    L.cae_door_destroy(None)  # a no-op


# ---- on the GPU --------------------------------------------------------------------------------------------------
@pytest.fixture(scope='module')
def cae(built_lib):
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
    assert door is not None and door.stats()['chunks'] == len(tiles)
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
def test_batch_limit_and_shapes_are_never_mixed(cae):
    """max_batch = 3, one in-flight slot: every batch holds at most 3 chunks of ONE shape; 40 calls from 12 threads."""
    from cnn_autoencoder_amd import synth
    from cnn_autoencoder_amd.frontdoor import FrontDoor
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3), seed=6)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    a, b = synth.uniform_tiles(24, 64), synth.uniform_tiles(16, 40, 72)
    ref = codec.encode_batch(a) + codec.encode_batch(b)
    tiles = list(a) + list(b)
    door = FrontDoor(codec, max_batch=3, inflight=1)
    order = np.random.default_rng(1).permutation(len(tiles))
    got = _run_threads(12, lambda i: (i, door.encode(tiles[i])), order)
    assert all(buf == ref[i] for i, buf in got)
    st = door.stats()
    assert st['chunks'] == 40 and st['batches'] >= 14  # ceil(24/3) + ceil(16/3)
    rec = _run_threads(12, lambda i: (i, door.decode(ref[i])), order)
    want = list(codec.decode_batch(ref[:24])) + list(codec.decode_batch(ref[24:]))
    assert all(np.array_equal(r, want[i]) for i, r in rec)
    # coalescing, deterministically: 10 calls of one shape queue up behind a held dispatcher and leave as 3 + 3 + 3 + 1
    import time
    door.stats(reset=True)
    door.hold(True)
    with ThreadPoolExecutor(10) as pool:
        futs = [pool.submit(door.encode, tiles[i]) for i in range(10)]
        for _ in range(2000):
            if door.stats()['queued'] == 10:
                break
            time.sleep(0.005)
        assert door.stats()['queued'] == 10 and door.stats()['batches'] == 0
        door.hold(False)
        assert [f.result(timeout=60) for f in futs] == ref[:10]
    st = door.stats()
    assert st['chunks'] == 10 and st['batches'] == 4
    # buffer-protocol inputs (zarr hands over whatever the store returned) and a header-only chunk
    assert np.array_equal(door.decode(bytearray(ref[0])), want[0])
    assert np.array_equal(door.decode(memoryview(ref[30])), want[30])
    with pytest.raises(Exception):
        door.decode(ref[0][:12])
    door.close()
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
    got = _run_threads(6, codec.encode, list(tiles))
    assert got == ref
    st = codec._front_door().stats()
    if enc.precision_code() == 1:
        assert st['fp32_repeats'] >= 1, 'the stress weights did not trip the guard'
    rec = _run_threads(6, codec.decode, ref)
    assert all(np.array_equal(r, w) for r, w in zip(rec, codec.decode_batch(ref)))
    codec.close()
