"""Zarr-v2 tile I/O: on-disk layout on CPU, codec flows on the GPU."""
import json
import os

import numpy as np
import pytest
import torch

from cnn_autoencoder_amd import zarrio


def test_zarr_v2_layout_and_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (150, 200, 3), dtype=np.uint8)
    store = str(tmp_path / 'img.zarr')
    z = zarrio.ZarrArray.create(store, '0/0', img.shape, (64, 64, 3), np.uint8, codec=zarrio.Zlib(level=9))
    z[:] = img
    # metadata exactly as zarr v2 writes it
    meta = json.load(open(os.path.join(store, '0', '0', '.zarray')))
    assert meta == dict(zarr_format=2, shape=[150, 200, 3], chunks=[64, 64, 3], dtype='|u1',
                        compressor={'id': 'zlib', 'level': 9}, fill_value=0, order='C', filters=None,
                        dimension_separator='.')
    for g in (store, os.path.join(store, '0')):
        assert json.load(open(os.path.join(g, '.zgroup'))) == {'zarr_format': 2}
    files = sorted(f for f in os.listdir(os.path.join(store, '0', '0')) if not f.startswith('.'))
    assert files == sorted(f'{i}.{j}.0' for i in range(3) for j in range(4))
    # edge chunks are stored at the full chunk shape, zero padded
    import zlib
    edge = np.frombuffer(zlib.decompress(open(os.path.join(store, '0', '0', '2.3.0'), 'rb').read()), dtype=np.uint8)
    edge = edge.reshape(64, 64, 3)
    assert np.array_equal(edge[:22, :8], img[128:, 192:]) and not edge[22:].any() and not edge[:, 8:].any()
    back = zarrio.ZarrArray.open(store, '0/0')
    assert back.shape == (150, 200, 3) and back.grid == (3, 4, 1)
    assert np.array_equal(back[:], img)
    assert back.chunk_indices()[:5] == [(0, 0, 0), (0, 1, 0), (0, 2, 0), (0, 3, 0), (1, 0, 0)]


def test_uncompressed_and_missing_chunks(tmp_path):
    store = str(tmp_path / 'raw.zarr')
    z = zarrio.ZarrArray.create(store, 'a', (5, 7), (4, 4), np.float32, codec=None, fill_value=0)
    data = np.arange(35, dtype=np.float32).reshape(5, 7)
    z.write_chunk((0, 0), data[:4, :4])
    z.write_chunk((1, 1), data[4:, 4:])
    got = zarrio.ZarrArray.open(store, 'a')[:]
    want = np.zeros_like(data)
    want[:4, :4] = data[:4, :4]
    want[4:, 4:] = data[4:, 4:]
    assert np.array_equal(got, want)
    assert json.load(open(os.path.join(store, 'a', '.zarray')))['compressor'] is None


def test_unknown_codecs_raise(tmp_path):
    with pytest.raises(ValueError, match='not supported'):
        zarrio.get_codec({'id': 'blosc', 'cname': 'zlib'})
    with pytest.raises(ValueError, match='not supported'):
        zarrio.compress_image('Jpeg2k', None, np.zeros((8, 8, 3), np.uint8), str(tmp_path / 'x.zarr'))
    with pytest.raises(ValueError):
        zarrio.compress_image('None', None, np.zeros((8, 8), np.uint8), str(tmp_path / 'y.zarr'))


@pytest.fixture
def checkpoint(tmp_path):
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    path = str(tmp_path / 'ckpt.pth')
    torch.save(synth.synthetic_state(cfg, seed=4), path)
    return path


@pytest.mark.gpu
def test_cae_codec_flow(tmp_path, checkpoint):
    """compress.py / decompress.py flow: zarr of 'cae' chunks written from an image with ragged edges,
    re-opened only from its metadata (Codec.from_config) and decoded."""
    import cnn_autoencoder_amd as cae
    from cnn_autoencoder_amd import synth
    img = synth.histo_tile(150, 3, 200)
    store = str(tmp_path / 'slide.zarr')
    z = zarrio.compress_image('CAE', checkpoint, img, store, patch_size=64, data_group='0/0')
    meta = json.load(open(os.path.join(store, '0', '0', '.zarray')))
    assert meta['compressor'] == {'id': 'cae', 'checkpoint': checkpoint, 'gpu': True}
    assert meta['chunks'] == [64, 64, 3] and meta['shape'] == [150, 200, 3]
    rec = zarrio.decompress_image(store, '0/0')
    assert rec.shape == img.shape and rec.dtype == np.uint8
    # chunk files are exactly codec.encode(padded chunk); decode of each equals the assembled image
    codec = cae.ConvolutionalAutoencoder(checkpoint=checkpoint)
    for idx in [(0, 0, 0), (2, 3, 0)]:
        raw = z.read_chunk_bytes(idx)
        padded = z.pad_chunk(img[z.chunk_slices(idx)])
        assert raw == codec.encode(padded)
        sl = z.chunk_slices(idx)
        assert np.array_equal(codec.decode(raw)[:sl[0].stop - sl[0].start, :sl[1].stop - sl[1].start], rec[sl])


@pytest.mark.gpu
def test_bottleneck_flow(tmp_path, checkpoint):
    """-sbn mode (compress.py:38-62,103-113): latents stored through 'cae_bn' with the entropy parameters in
    the zarr metadata; decompress.py:61-79 runs the decoder.  Same pixels as the direct 'cae' flow."""
    from cnn_autoencoder_amd import synth
    img = synth.histo_tile(128, 5, 192)
    direct = str(tmp_path / 'direct.zarr')
    bn = str(tmp_path / 'bn.zarr')
    zarrio.compress_image('CAE', checkpoint, img, direct, patch_size=64)
    z = zarrio.compress_image('CAE', checkpoint, img, bn, patch_size=64, save_as_bottleneck=True)
    meta = json.load(open(os.path.join(bn, '0', '0', '.zarray')))
    assert meta['compressor']['id'] == 'cae_bn' and meta['dtype'] == '<f4'
    assert set(meta['compressor']) == {'id', 'filters', 'channels_bn', 'fact_ent_checkpoint'}
    assert meta['chunks'] == [8, 8, 48] and meta['shape'] == [16, 24, 48]
    a = zarrio.decompress_image(direct)
    b = zarrio.decompress_image(bn, checkpoint=checkpoint)
    assert a.shape == (128, 192, 3) and b.shape == (128, 192, 3)
    assert np.array_equal(a, b)
    # the direct and the bottleneck stores hold the same entropy-coded payloads, headers aside
    for idx in z.chunk_indices():
        d = open(os.path.join(direct, '0', '0', '.'.join(map(str, idx))), 'rb').read()
        assert z.read_chunk_bytes(idx)[16:] == d[16:]
