"""Zarr-v2 tile I/O and the compress / decompress flows of the reference, without third-party deps.

The reference delegates tile I/O to dask + zarr (``src/compress.py:78-128``,
``src/decompress.py:48-96``): the image is rechunked to ``(patch, patch, 3)`` and written with
``compressor=codec``, so every chunk file is exactly ``codec.encode(chunk)``.  This module writes and
reads the same on-disk layout (zarr format 2, directory store):

    <store>/<group>/.zgroup                       {"zarr_format": 2}
    <store>/<group>/<array>/.zarray               shape, chunks, dtype, compressor config, fill_value, order "C",
                                                  filters null, dimension_separator "."
    <store>/<group>/<array>/<i>.<j>.<k>           one file per chunk = the codec's bytes of the FULL chunk shape
                                                  (edge chunks are padded with fill_value, as zarr does)

so an array written here opens with ``zarr.open`` once ``register_codecs()`` has run, and vice versa.
Chunks are coded in batches on the GPU (``encode_batch`` / ``decode_batch``) instead of one dask task per
chunk; with ``torch.distributed`` initialised every rank codes the contiguous tile block
``slide.tile_range(rank, world, n_tiles)`` and rank 0 writes the metadata.
"""
from __future__ import annotations

import json
import math
import os
import zlib
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

ZARR_FORMAT = 2


class _Raw:
    codec_id = None

    def encode(self, buf):
        return np.ascontiguousarray(buf).tobytes()

    def decode(self, buf, out=None):
        return np.frombuffer(bytes(buf), dtype=np.uint8)

    def get_config(self):
        return None


class Zlib:
    """numcodecs.Zlib-compatible codec (the reference writes decompressed images with Blosc/zlib-9,
    decompress.py:48; Blosc itself is a third-party library and is not reimplemented)."""
    codec_id = 'zlib'

    def __init__(self, level: int = 1):
        self.level = int(level)

    def encode(self, buf):
        return zlib.compress(np.ascontiguousarray(buf).tobytes(), self.level)

    def decode(self, buf, out=None):
        return np.frombuffer(zlib.decompress(bytes(buf)), dtype=np.uint8)

    def get_config(self):
        return dict(id=self.codec_id, level=self.level)

    @classmethod
    def from_config(cls, config):
        config = dict(config)
        config.pop('id', None)
        return cls(**config)


def get_codec(config: Optional[dict]):
    """compressor config of a .zarray -> codec object (numcodecs.get_codec for the ids this path uses)."""
    if config is None:
        return _Raw()
    cid = config.get('id')
    if cid == 'zlib':
        return Zlib.from_config(config)
    if cid in ('cae', 'cae_bn'):
        from .codec import ConvolutionalAutoencoder, ConvolutionalAutoencoderBottleneck
        cls = ConvolutionalAutoencoder if cid == 'cae' else ConvolutionalAutoencoderBottleneck
        return cls.from_config(config)
    raise ValueError(f'Codec {cid!r} not supported')


def _ensure_groups(store: str, path: str):
    os.makedirs(store, exist_ok=True)
    parts = [p for p in path.split('/') if p]
    cur = store
    for p in [None] + parts[:-1]:
        if p is not None:
            cur = os.path.join(cur, p)
            os.makedirs(cur, exist_ok=True)
        zg = os.path.join(cur, '.zgroup')
        if not os.path.exists(zg) and not os.path.exists(os.path.join(cur, '.zarray')):
            with open(zg, 'w') as f:
                json.dump({'zarr_format': ZARR_FORMAT}, f)


class ZarrArray:
    """One zarr-v2 array in a directory store."""

    def __init__(self, root: str, meta: dict, codec=None):
        self.root = root
        self.meta = meta
        self.shape = tuple(meta['shape'])
        self.chunks = tuple(meta['chunks'])
        self.dtype = np.dtype(meta['dtype'])
        self.sep = meta.get('dimension_separator', '.')
        self.fill_value = meta.get('fill_value', 0) or 0
        self.codec = codec if codec is not None else get_codec(meta.get('compressor'))

    # ---- construction ------------------------------------------------------------------------
    @classmethod
    def create(cls, store: str, component: str, shape, chunks, dtype, codec=None, fill_value=0,
               write_meta: bool = True) -> 'ZarrArray':
        root = os.path.join(store, component) if component else store
        compressor = codec.get_config() if codec is not None else None
        meta = dict(zarr_format=ZARR_FORMAT, shape=[int(s) for s in shape], chunks=[int(c) for c in chunks],
                    dtype=np.dtype(dtype).str, compressor=compressor, fill_value=fill_value, order='C',
                    filters=None, dimension_separator='.')
        try:
            text = json.dumps(meta, indent=4, sort_keys=True)
        except TypeError as e:  # e.g. a 'cae' codec built from an in-memory checkpoint dict
            raise ValueError(f'the codec configuration cannot be stored in .zarray metadata ({e}): build the codec '
                             'from a checkpoint PATH when writing a store') from e
        # every rank creates the directories it writes chunk files into (a rank may finish its first batch before the
        # metadata writer has run); only the metadata writer touches the .zgroup / .zarray files
        if root is not None:
            os.makedirs(root, exist_ok=True)
        if write_meta:
            _ensure_groups(store, component)
            with open(os.path.join(root, '.zarray'), 'w') as f:
                f.write(text)
        return cls(root, meta, codec=codec if codec is not None else _Raw())

    @classmethod
    def open(cls, store: str, component: str = '', codec=None) -> 'ZarrArray':
        root = os.path.join(store, component) if component else store
        with open(os.path.join(root, '.zarray')) as f:
            meta = json.load(f)
        if meta.get('zarr_format') != ZARR_FORMAT:
            raise ValueError('only zarr format 2 is supported')
        if meta.get('filters'):
            raise ValueError('filters are not supported')
        if meta.get('order', 'C') != 'C':
            raise ValueError('only C order is supported')
        return cls(root, meta, codec=codec)

    # ---- chunk grid --------------------------------------------------------------------------
    @property
    def grid(self) -> Tuple[int, ...]:
        return tuple(int(math.ceil(s / c)) for s, c in zip(self.shape, self.chunks))

    def chunk_indices(self) -> List[Tuple[int, ...]]:
        """All chunk coordinates in C (raster) order -- the tile order the slide driver shards."""
        return [tuple(int(v) for v in idx) for idx in np.ndindex(*self.grid)]

    def chunk_path(self, idx: Sequence[int]) -> str:
        return os.path.join(self.root, self.sep.join(str(int(i)) for i in idx))

    def chunk_slices(self, idx: Sequence[int]):
        return tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, self.chunks, self.shape))

    # ---- chunk I/O ---------------------------------------------------------------------------
    def pad_chunk(self, data: np.ndarray) -> np.ndarray:
        """Edge chunks are stored at the full chunk shape, padded with fill_value (zarr v2)."""
        if tuple(data.shape) == self.chunks:
            return np.ascontiguousarray(data, dtype=self.dtype)
        full = np.full(self.chunks, self.fill_value, dtype=self.dtype)
        full[tuple(slice(0, s) for s in data.shape)] = data
        return full

    def write_chunk_bytes(self, idx, cdata: bytes):
        with open(self.chunk_path(idx), 'wb') as f:
            f.write(cdata)

    def read_chunk_bytes(self, idx) -> Optional[bytes]:
        p = self.chunk_path(idx)
        if not os.path.exists(p):
            return None
        with open(p, 'rb') as f:
            return f.read()

    def write_chunk(self, idx, data: np.ndarray):
        self.write_chunk_bytes(idx, self.codec.encode(self.pad_chunk(data)))

    def read_chunk(self, idx) -> np.ndarray:
        cdata = self.read_chunk_bytes(idx)
        if cdata is None:
            return np.full(self.chunks, self.fill_value, dtype=self.dtype)
        out = np.asarray(self.codec.decode(cdata))
        return np.ascontiguousarray(out).view(self.dtype).reshape(self.chunks) if out.dtype != self.dtype \
            else out.reshape(self.chunks)

    def __setitem__(self, key, value):
        if key != slice(None) and key is not Ellipsis:
            raise NotImplementedError('only whole-array assignment is supported')
        value = np.asarray(value)
        if tuple(value.shape) != self.shape:
            raise ValueError(f'shape mismatch {value.shape} vs {self.shape}')
        for idx in self.chunk_indices():
            self.write_chunk(idx, value[self.chunk_slices(idx)])

    def __getitem__(self, key) -> np.ndarray:
        if key != slice(None) and key is not Ellipsis:
            raise NotImplementedError('only whole-array reads are supported')
        out = np.empty(self.shape, dtype=self.dtype)
        for idx in self.chunk_indices():
            sl = self.chunk_slices(idx)
            chunk = self.read_chunk(idx)
            out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
        return out


# ---- the reference's compress / decompress flows ---------------------------------------------------

def _rank_world():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank(), dist.get_world_size()
    except Exception:
        pass
    return 0, 1


def _barrier():
    """All ranks have written their chunk files (the store is complete when compress_image returns)."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
    except ImportError:
        pass


def _batches(items: Sequence, n: int) -> Iterable[Sequence]:
    for i in range(0, len(items), n):
        yield items[i:i + n]


def compress_image(codec: str, checkpoint, image: np.ndarray, output_filename: str, patch_size: int = 512,
                   data_group: str = '0/0', save_as_bottleneck: bool = False, gpu: bool = True,
                   batch_tiles: int = 32) -> ZarrArray:
    """compress.py:29-128 for an in-memory (H, W, C) uint8 image: rechunk to (patch, patch, C) and write a
    zarr array whose chunks are ``codec`` bitstreams.  codec: 'CAE' | 'Zlib' | 'None' (the reference's
    'Blosc' / 'Jpeg*' names are other libraries' codecs and raise ValueError here)."""
    from . import slide
    from .codec import (ConvolutionalAutoencoder, ConvolutionalAutoencoderBottleneck, _module,
                        autoencoder_from_state_dict)
    image = np.asarray(image)
    if image.ndim != 3 or image.dtype != np.uint8:
        raise ValueError(f'expected a (H, W, C) uint8 image, got {image.dtype} {image.shape}')
    if not len(data_group):
        data_group = '0/0'
    h, w, c = image.shape
    rank, world = _rank_world()
    if 'CAE' in codec and not save_as_bottleneck and not isinstance(checkpoint, str):
        # the 'cae' codec persists its checkpoint PATH in the .zarray metadata (_autoencoders.py:532): fail before
        # the model is built, not when the metadata is written
        raise ValueError("codec 'CAE' stores the checkpoint path in the array metadata: pass a path, not a dict")

    if 'CAE' in codec and save_as_bottleneck:
        import torch
        model = autoencoder_from_state_dict(checkpoint=checkpoint, gpu=gpu, train=False)
        fe = _module(model['fact_ent'])
        enc = _module(model['encoder'])
        level = len(enc.analysis_track)
        compressor = ConvolutionalAutoencoderBottleneck(channels_bn=fe.channels, fact_ent=fe, gpu=gpu)
        lat = int(math.ceil(patch_size / 2 ** level))
        gy, gx = int(math.ceil(h / patch_size)), int(math.ceil(w / patch_size))
        # latent array: per tile ceil(patch / 2^L) rows/cols (compress.py:103-107)
        z = ZarrArray.create(output_filename, data_group, (gy * lat, gx * lat, fe.channels), (lat, lat, fe.channels),
                             np.float32, codec=compressor, write_meta=rank == 0)
        src = ZarrArray(None, dict(shape=[h, w, c], chunks=[patch_size, patch_size, c], dtype='|u1'), codec=_Raw())
        tiles = src.chunk_indices()
        lo, hi = slide.tile_range(rank, world, len(tiles))
        for group in _batches(tiles[lo:hi], batch_tiles):
            batch = np.stack([src.pad_chunk(image[src.chunk_slices(i)]) for i in group])
            with torch.no_grad():
                y = enc.forward_u8(torch.from_numpy(batch).cuda())
                strings = fe.compress(y)
            for idx, s in zip(group, strings):
                import struct
                z.write_chunk_bytes(idx, struct.pack('>QQ', y.shape[2], y.shape[3]) + s)
        _barrier()
        return z

    if 'CAE' in codec:
        compressor = ConvolutionalAutoencoder(checkpoint=checkpoint, gpu=gpu)
    elif 'Zlib' in codec:
        compressor = Zlib(level=9)
    elif 'None' in codec:
        compressor = None
    else:
        raise ValueError('Codec %s not supported' % codec)
    z = ZarrArray.create(output_filename, data_group, (h, w, c), (patch_size, patch_size, c), np.uint8,
                         codec=compressor, write_meta=rank == 0)
    tiles = z.chunk_indices()
    lo, hi = slide.tile_range(rank, world, len(tiles))
    if isinstance(compressor, ConvolutionalAutoencoder):
        # pipelined: the GPU analyses the next batches while a host worker range-encodes and this thread writes files
        import struct
        groups = list(_batches(tiles[lo:hi], batch_tiles))
        coder = slide.SlideCoder(compressor)
        stream = coder.compress_batches(np.stack([z.pad_chunk(image[z.chunk_slices(i)]) for i in g]) for g in groups)
        head = struct.pack('>QQ', patch_size, patch_size)  # chunk = '>QQ'(h, w) + rANS payload (_autoencoders.py:553-555)
        for group, payloads in zip(groups, stream):
            for idx, payload in zip(group, payloads):
                z.write_chunk_bytes(idx, head + payload)
        _barrier()
        return z
    for idx in tiles[lo:hi]:
        z.write_chunk(idx, image[z.chunk_slices(idx)])
    _barrier()
    return z


def decompress_image(input_filename: str, data_group: str = '0/0', checkpoint=None, gpu: bool = True,
                     batch_tiles: int = 32) -> np.ndarray:
    """decompress.py:40-96: open the zarr (chunk decode = the stored codec), and when `checkpoint` is given the
    array holds 'cae_bn' latents that the decoder turns back into pixels (decompress.py:61-79)."""
    from .codec import ConvolutionalAutoencoder, _module, autoencoder_from_state_dict
    z = ZarrArray.open(input_filename, data_group)
    if checkpoint is None or (isinstance(checkpoint, str) and not len(checkpoint)):
        if isinstance(z.codec, ConvolutionalAutoencoder):
            # pipelined: a host worker range-decodes the next batches while the GPU synthesises
            import struct
            from . import slide
            out = np.empty(z.shape, dtype=z.dtype)
            groups = list(_batches(z.chunk_indices(), batch_tiles))
            ph, pw = z.chunks[0], z.chunks[1]

            def payloads(group):
                bufs = [z.read_chunk_bytes(i) for i in group]
                if any(b is None for b in bufs):
                    raise ValueError('missing chunk file')
                for b in bufs:
                    if struct.unpack('>QQ', b[:16]) != (ph, pw):
                        raise ValueError('chunk header does not match the chunk shape')
                return [b[16:] for b in bufs]

            coder = slide.SlideCoder(z.codec)
            stream = coder.decompress_batches((payloads(g) for g in groups), ph, pw, to_host=True)
            for group, rec in zip(groups, stream):  # rec: pinned ring buffer, copied out right away
                for idx, chunk in zip(group, rec):
                    sl = z.chunk_slices(idx)
                    out[sl] = chunk[tuple(slice(0, s.stop - s.start) for s in sl)]
            return out
        return z[:]
    import torch
    model = autoencoder_from_state_dict(checkpoint=checkpoint, gpu=gpu, train=False)
    dec = _module(model['decoder'])
    scale = 2 ** dec.rec_level
    lat = z[:]  # (gy*lat, gx*lat, C) float32 latents, decoded by the 'cae_bn' codec
    ly, lx = z.chunks[0], z.chunks[1]
    out = np.empty((lat.shape[0] * scale, lat.shape[1] * scale, dec._dims[0]), dtype=np.uint8)
    idxs = z.chunk_indices()
    for group in _batches(idxs, batch_tiles):
        batch = np.stack([lat[i * ly:(i + 1) * ly, j * lx:(j + 1) * lx] for i, j, _ in group])
        with torch.no_grad():
            rec = dec.forward_u8(torch.from_numpy(batch).permute(0, 3, 1, 2).contiguous().cuda()).cpu().numpy()
        for (i, j, _), tile in zip(group, rec):
            out[i * ly * scale:(i + 1) * ly * scale, j * lx * scale:(j + 1) * lx * scale] = tile
    return out
