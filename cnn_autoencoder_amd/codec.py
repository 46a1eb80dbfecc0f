"""Model factory and numcodecs codecs with the reference's plugin surface.

Mirrors ``src/models/tasks/_autoencoders.py``: ``setup_modules`` :458-479,
``load_state_dict`` :482-502, ``autoencoder_from_state_dict`` :505-527,
``ConvolutionalAutoencoder`` (codec_id 'cae') :530-584 and
``ConvolutionalAutoencoderBottleneck`` ('cae_bn') :587-673 -- same constructor kwargs
(so ``Codec.from_config`` round-trips through ``.zarray`` metadata), same chunk byte format
(16-byte ``>QQ`` header + rANS payload), same return types.  ``encode``/``decode`` keep the
one-tile-per-call contract zarr uses; ``encode_batch``/``decode_batch`` are the batched side
doors the sharded slide driver uses.
"""
from __future__ import annotations

import base64
import io
import struct
import threading
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .entropy import EntropyBottleneck
from .modules import Analyzer, Synthesizer

try:  # numcodecs is an optional dependency of the *caller* (zarr); absent in the build image
    from numcodecs.abc import Codec as _CodecBase
    from numcodecs.compat import ensure_contiguous_ndarray, ndarray_copy
except Exception:  # pragma: no cover - exercised in the build image
    class _CodecBase:  # the part of numcodecs.abc.Codec the reference relies on
        codec_id = None

        def get_config(self):
            config = dict(id=self.codec_id)
            for k, v in self.__dict__.items():
                if not k.startswith('_'):
                    config[k] = v
            return config

        @classmethod
        def from_config(cls, config):
            config = dict(config)
            config.pop('id', None)
            return cls(**config)

        def __repr__(self):
            args = ', '.join(f'{k}={v!r}' for k, v in self.get_config().items() if k != 'id')
            return f'{type(self).__name__}({args})'

    def ensure_contiguous_ndarray(buf):
        return np.ascontiguousarray(np.asarray(buf))

    def ndarray_copy(src, dst):
        if dst is None:
            return src
        dst = np.asarray(dst)
        src = np.asarray(src)
        np.copyto(dst.reshape(-1).view(np.uint8), src.reshape(-1).view(np.uint8))
        return dst


def setup_modules(channels_bn=192, compression_level=4, K=4, r=3, enabled_modules=None, **kwargs) -> Dict[str, nn.Module]:
    if enabled_modules is None:
        enabled_modules = ['encoder', 'decoder', 'fact_ent']
    model: Dict[str, nn.Module] = {}
    if 'encoder' in enabled_modules:
        model['encoder'] = Analyzer(channels_bn=channels_bn, compression_level=compression_level, **kwargs)
    if 'decoder' in enabled_modules:
        model['decoder'] = Synthesizer(channels_bn=channels_bn, compression_level=compression_level, **kwargs)
    if 'fact_ent' in enabled_modules:
        model['fact_ent'] = EntropyBottleneck(channels=channels_bn, filters=[r] * K)
    return model


def load_state_dict(model, encoder=None, decoder=None, fact_ent=None, **kwargs):
    if 'encoder' in model and encoder is not None:
        model['encoder'].load_state_dict(encoder, strict=False)
    if 'decoder' in model and decoder is not None:
        model['decoder'].load_state_dict(decoder, strict=False)
    if 'fact_ent' in model and fact_ent is not None:
        # strict like the reference, except that a hand-built dict may omit the derived buffers
        have = set(fact_ent.keys())
        derived = {'_offset', '_quantized_cdf', '_cdf_length', 'target', 'likelihood_lower_bound.bound'}
        strict = derived <= have
        model['fact_ent'].load_state_dict(fact_ent, strict=strict)
        missing = [k for k in model['fact_ent'].state_dict() if k not in have and k not in derived]
        if missing:
            raise RuntimeError(f'Missing key(s) in fact_ent state_dict: {missing}')
        model['fact_ent'].update(force=True)


def _load_checkpoint(path: str):
    """torch.load restricted to tensors and plain containers.  The path of a 'cae' codec comes out of a .zarray file
    (Codec.from_config), i.e. from whoever wrote the store: a pickle that needs more than the reference's schema
    (numbers, strings, lists, dicts, tensors) is refused unless CAE_TRUSTED_CHECKPOINT=1 says the file is
    trusted."""
    import os
    import pickle
    try:
        return torch.load(path, map_location='cpu', weights_only=True)
    except pickle.UnpicklingError as e:
        if os.environ.get('CAE_TRUSTED_CHECKPOINT') == '1':
            return torch.load(path, map_location='cpu', weights_only=False)
        raise ValueError(f'checkpoint {path!r} holds objects beyond tensors and plain containers ({e}); set '
                         'CAE_TRUSTED_CHECKPOINT=1 to unpickle a file you trust') from e


def autoencoder_from_state_dict(checkpoint, gpu=False, train=False) -> Dict[str, nn.Module]:
    """checkpoint: path to a ``torch.save``d dict or the dict itself (SURVEY §3.5 schema).

    ``gpu`` is accepted for signature compatibility; the modules always execute on the current
    HIP device (there is no CPU path), one process per GPU.
    """
    if isinstance(checkpoint, str):
        state = _load_checkpoint(checkpoint)
    else:
        state = checkpoint
    model = setup_modules(**state)
    load_state_dict(model, **state)
    use_cuda = torch.cuda.is_available()
    for k in list(model.keys()):
        if use_cuda:
            model[k].cuda()
            model[k] = nn.DataParallel(model[k], device_ids=[torch.cuda.current_device()])
        else:
            model[k] = nn.DataParallel(model[k])  # no devices: forward calls .module directly
        model[k].train() if train else model[k].eval()
    return model


def _module(m):
    return m.module if isinstance(m, nn.DataParallel) else m


class ConvolutionalAutoencoder(_CodecBase):
    codec_id = 'cae'

    def __init__(self, checkpoint, gpu=False):
        self.checkpoint = checkpoint
        self.gpu = gpu
        self._model = autoencoder_from_state_dict(checkpoint, gpu=gpu, train=False)
        # dask calls encode / decode from a thread pool on this one instance (compress.py:121-128).  The lock covers
        # the GPU section only (the tracks' workspaces are used in stream order); range coding runs outside it.
        self._lock = threading.Lock()
        self._door = None
        self._door_mu = threading.Lock()

    def _front_door(self):
        """The micro-batching front door of encode() / decode() (frontdoor.FrontDoor), created on first use;
        CAE_DOOR=0 keeps every call a batch of one."""
        import os
        if os.environ.get('CAE_DOOR', '1') == '0':
            return None
        if self._door is None:
            with self._door_mu:
                if self._door is None:
                    from .frontdoor import FrontDoor
                    self._door = FrontDoor(self)
        return self._door

    def close(self):
        """Stops the front door's service threads (they are daemons: optional)."""
        if self._door is not None:
            self._door.close()
            self._door = None

    # ---- batched side doors ------------------------------------------------------------------
    @torch.no_grad()
    def encode_batch(self, tiles: np.ndarray) -> List[bytes]:
        """tiles (n,h,w,c) uint8 -> n chunk byte strings."""
        tiles = np.ascontiguousarray(tiles)
        if tiles.ndim != 4 or tiles.dtype != np.uint8:
            raise ValueError(f'expected uint8 (n,h,w,c), got {tiles.dtype} {tiles.shape}')
        n, h, w, _ = tiles.shape
        dev = _lib.require_gpu()
        eb = _module(self._model['fact_ent'])
        with self._lock:  # GPU section: H2D, analysis with the fused quantiser, D2H of the symbols
            x = torch.from_numpy(tiles).to(dev)
            sym = _module(self._model['encoder']).forward_u8_symbols(x, eb)
            sym_host = sym.reshape(n, sym.size(1), -1).cpu().numpy()
        strings = eb.encode_symbols(sym_host)  # host range coder: outside the lock
        head = struct.pack('>QQ', h, w)
        return [head + s for s in strings]

    @torch.no_grad()
    def decode_batch(self, bufs: Sequence[bytes]) -> np.ndarray:
        """chunk byte strings of equal tile size -> (n,h,w,c) uint8."""
        dec = _module(self._model['decoder'])
        eb = _module(self._model['fact_ent'])
        level = len(dec.synthesis_track)
        hw = {struct.unpack('>QQ', bytes(b[:16])) for b in bufs}
        if len(hw) != 1:
            raise ValueError('decode_batch needs chunks of one tile size')
        h, w = hw.pop()
        lh, lw = h // 2 ** level, w // 2 ** level
        dev = _lib.require_gpu()
        sym_host = eb.decode_symbols([bytes(b[16:]) for b in bufs], lh * lw)  # host range decoder: outside the lock
        with self._lock:
            sym = torch.from_numpy(sym_host).to(dev).reshape(len(bufs), eb.channels, lh, lw)
            out = dec.forward_symbols_u8(sym, eb)
            return out.cpu().numpy()

    # ---- numcodecs contract ----------------------------------------------------------------------
    def encode(self, buf):
        buf = np.asarray(buf)
        if buf.ndim != 3:
            raise ValueError(f'expected an (h,w,c) chunk, got shape {buf.shape}')
        door = self._front_door()
        if door is not None:
            return door.encode(buf)
        return self.encode_batch(buf[None])[0]

    def decode(self, buf, out=None):
        if out is not None:
            out = ensure_contiguous_ndarray(out)
        door = self._front_door()
        if door is not None:
            x_r = door.decode(buf, out)
            if x_r is out:
                return out
        else:
            x_r = np.ascontiguousarray(self.decode_batch([bytes(buf)])[0])
        return ndarray_copy(ensure_contiguous_ndarray(x_r), out)


class ConvolutionalAutoencoderBottleneck(_CodecBase):
    codec_id = 'cae_bn'

    def __init__(self, channels_bn, fact_ent=None, filters=None, fact_ent_checkpoint=None, gpu=False):
        if fact_ent is not None:
            filters = list(fact_ent.filters)
            fact_ent_checkpoint = {}
            for n, par in fact_ent.named_parameters():
                fact_ent_checkpoint[n] = self._tensor2bytes(par)
        self.filters = filters
        self.channels_bn = channels_bn
        self.fact_ent_checkpoint = fact_ent_checkpoint
        self._lock = threading.Lock()
        self._setup_encoder(gpu)

    def _setup_encoder(self, gpu=False):
        self._fact_ent = EntropyBottleneck(channels=self.channels_bn, filters=self.filters)
        state = {n: self._bytes2tensor(par) for n, par in self.fact_ent_checkpoint.items()}
        self._fact_ent.load_state_dict(state, strict=False)
        self._fact_ent.update(force=True)
        self._fact_ent.eval()
        if torch.cuda.is_available():
            self._fact_ent.cuda()

    @staticmethod
    def _tensor2bytes(tensor):
        buf = io.BytesIO()
        torch.save(tensor.cpu().detach(), buf)
        return base64.b64encode(buf.getvalue()).decode('ascii')

    @staticmethod
    def _bytes2tensor(buf):
        # the blob arrives through zarr metadata (Codec.from_config): tensors only, never a general unpickle
        import pickle
        try:
            t = torch.load(io.BytesIO(base64.b64decode(buf)), weights_only=True)
        except pickle.UnpicklingError as e:
            raise ValueError(f'fact_ent_checkpoint entry is not a plain tensor: {e}') from e
        if not isinstance(t, torch.Tensor):
            raise ValueError(f'fact_ent_checkpoint entry is a {type(t).__name__}, expected a tensor')
        return t

    @torch.no_grad()
    def encode(self, buf):
        buf = np.ascontiguousarray(buf)
        if buf.ndim != 3:
            raise ValueError(f'expected an (h,w,c) latent chunk, got shape {buf.shape}')
        h, w, c = buf.shape
        y = torch.from_numpy(buf).permute(2, 0, 1).reshape(1, c, h, w)
        with self._lock:  # GPU section (quantiser kernel); the range coder runs outside, in the caller's thread
            sym = self._fact_ent.quantize_symbols(y)
            sym_host = sym.reshape(1, c, -1).cpu().numpy()
        s = self._fact_ent.encode_symbols(sym_host, threads=1)
        return struct.pack('>QQ', h, w) + s[0]

    @torch.no_grad()
    def decode(self, buf, out=None):
        if out is not None:
            out = ensure_contiguous_ndarray(out)
        buf = bytes(buf)
        h, w = struct.unpack('>QQ', buf[:16])
        sym_host = self._fact_ent.decode_symbols([buf[16:]], h * w, threads=1)  # range decoder: outside the lock
        with self._lock:
            sym = torch.from_numpy(sym_host).reshape(1, self.channels_bn, h, w)
            y_q = self._fact_ent.dequantize_symbols(sym).cpu()
        y_q = np.ascontiguousarray(y_q[0].permute(1, 2, 0).float().numpy())
        return ndarray_copy(ensure_contiguous_ndarray(y_q), out)


def register_codecs():
    """numcodecs.register_codec for both codecs (compress.py:25-26) when numcodecs is installed."""
    import numcodecs
    numcodecs.register_codec(ConvolutionalAutoencoder)
    numcodecs.register_codec(ConvolutionalAutoencoderBottleneck)
