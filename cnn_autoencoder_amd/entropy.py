"""Factorized entropy bottleneck: host-side mirror of ``compressai.entropy_models.EntropyBottleneck``.

The reference builds this class at ``src/models/tasks/_autoencoders.py:476-477`` and calls
``update(force=True)`` (:502), ``compress`` (:549-551), ``decompress`` (:568-571), ``loss()``
(``criteria/_lossutils.py:70``) and ``__call__`` (``_taskutils.py:97``) on it.  Same constructor,
parameter / buffer names and return conventions here; the quantiser runs as a HIP kernel and the
range coder / CDF quantiser are the C++ entry points of libcae_hip.so (no compressai, no CPU
fallback for the coding path).
"""
from __future__ import annotations

import ctypes
import threading
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


_SYNC_MU = threading.Lock()  # first use of a module's coder tables from several threads at once (codec front door)


class _LowerBoundFn(torch.autograd.Function):
    """max(x, bound) with the compressai gradient rule (pass where x >= bound or grad < 0)."""

    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        pass_through = (x >= bound) | (g < 0)
        return pass_through.to(g.dtype) * g, None


class LowerBound(nn.Module):
    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer('bound', torch.Tensor([float(bound)]))

    def forward(self, x):
        return _LowerBoundFn.apply(x, self.bound)


class PackedStreams:
    """The rANS streams of one batch in a single library-owned buffer (cae_rans_encode_packed): a read-only sequence of
    byte strings that is handed from the encoder to the decoder / file writer without per-stream copies.
    ``len(p)``, ``p.nbytes(i)``, ``p[i]`` (a ``bytes`` copy), iteration, ``==`` with a list of bytes."""

    def __init__(self, ptr: int, offsets):
        self._ptr = ptr
        self.offsets = [int(o) for o in offsets]

    def __len__(self):
        return len(self.offsets) - 1

    def nbytes(self, i: int) -> int:
        return self.offsets[i + 1] - self.offsets[i]

    def pointer(self, i: int) -> int:
        return self._ptr + self.offsets[i]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        return ctypes.string_at(self.pointer(i), self.nbytes(i))

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __eq__(self, other):
        try:
            return len(other) == len(self) and all(bytes(a) == bytes(b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def __del__(self):
        if getattr(self, '_ptr', None):
            try:
                _lib.lib().cae_free(self._ptr)
            except Exception:
                pass
            self._ptr = None


def pmf_to_quantized_cdf(pmf: torch.Tensor, precision: int = 16) -> torch.Tensor:
    """compressai._CXX.pmf_to_quantized_cdf through the C ABI (cae_pmf_to_quantized_cdf)."""
    p = np.ascontiguousarray(pmf.detach().cpu().numpy(), dtype=np.float32)
    out = np.empty(p.shape[0] + 1, dtype=np.uint32)
    _lib.check(_lib.lib().cae_pmf_to_quantized_cdf(p.ctypes.data, p.shape[0], precision, out.ctypes.data))
    return torch.from_numpy(out.astype(np.int32))


def _logits_cumulative(params, n_filters: int, inputs: torch.Tensor) -> torch.Tensor:
    """Cumulative-logit network of the factorized prior."""
    logits = inputs
    for i in range(n_filters + 1):
        logits = torch.matmul(F.softplus(params[f'_matrix{i:d}']), logits)
        logits = logits + params[f'_bias{i:d}']
        if i < n_filters:
            logits = logits + torch.tanh(params[f'_factor{i:d}']) * torch.tanh(logits)
    return logits


# Floating-point form of p = c(y + 1/2) - c(y - 1/2).  'plain' (default): sigmoid(u) - sigmoid(l), what compressai
# >= 1.2.x is believed to compute (its `_likelihood` returns (likelihood, lower, upper) as this code assumes; the
# reference requires compressai >= 1.2.4, requirements.txt:25).  'sign_trick': |sigmoid(s u) - sigmoid(s l)| with
# s = -sign(l + u), the older compressai / tensorflow-compression form.  Equal in exact arithmetic; in fp32 the CDF
# tables (hence bitstreams) can differ by one unit in the upper tail (tests/test_host.py shows where).  compressai
# is absent from the build image, so the choice cannot be pinned: it is a named, switchable option
# (EntropyBottleneck(likelihood_form=...) or CAE_LIKELIHOOD_FORM).
LIKELIHOOD_FORMS = ('plain', 'sign_trick')


def _likelihood(params, n_filters: int, inputs: torch.Tensor, form: str = 'plain'):
    lower = _logits_cumulative(params, n_filters, inputs - 0.5)
    upper = _logits_cumulative(params, n_filters, inputs + 0.5)
    if form == 'plain':
        likelihood = torch.sigmoid(upper) - torch.sigmoid(lower)
    else:
        sign = -torch.sign(lower + upper).detach()
        likelihood = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
    return likelihood, lower, upper


class _DensityTrainFn(torch.autograd.Function):
    """y (N, C, ...) + noise -> (y~, max(p(y~), bound)) with hand-written forward and backward kernels
    (csrc/cae_density_train.hip; reference: compressai EntropyBottleneck.forward(training=True), _autoencoders.py:502).
    `params`: matrices 0..K, biases 0..K, factors 0..K-1 in their stored form; gradients come back for each of them."""

    @staticmethod
    def forward(ctx, y, noise, plain, bound, *params):
        L = _lib.lib()
        y = y.contiguous().float()
        n, c = y.size(0), y.size(1)
        hw = y.numel() // (n * c)
        raw = torch.cat([p.detach().reshape(c, -1).float() for p in params], dim=1).contiguous()
        out, lik = torch.empty_like(y), torch.empty_like(y)
        noise_c = None if noise is None else noise.to(y).contiguous()
        _lib.check(L.cae_t_density_forward(y.data_ptr(), None if noise_c is None else noise_c.data_ptr(), raw.data_ptr(),
                                           n, c, hw, int(plain), float(bound), out.data_ptr(), lik.data_ptr(),
                                           _lib.stream_ptr()))
        ctx.save_for_backward(out, raw)
        ctx.meta = (n, c, hw, int(plain), float(bound), [tuple(p.shape) for p in params])
        return out, lik

    @staticmethod
    def backward(ctx, g_out, g_lik):
        L = _lib.lib()
        out, raw = ctx.saved_tensors
        n, c, hw, plain, bound, shapes = ctx.meta
        g_lik = torch.zeros_like(out) if g_lik is None else g_lik.contiguous().float()
        g_out_c = None if g_out is None else g_out.contiguous().float()
        g_y, g_raw = torch.empty_like(out), torch.empty_like(raw)
        _lib.check(L.cae_t_density_backward(out.data_ptr(), g_lik.data_ptr(), None if g_out_c is None else g_out_c.data_ptr(),
                                            raw.data_ptr(), n, c, hw, plain, bound, g_y.data_ptr(), g_raw.data_ptr(),
                                            _lib.stream_ptr()))
        grads, col = [], 0
        for shp in shapes:  # views into the (C, NP) gradient block
            k = int(np.prod(shp[1:]))
            grads.append(g_raw[:, col:col + k].reshape(shp))
            col += k
        return (g_y, None, None, None, *grads)


class EntropyBottleneck(nn.Module):
    def __init__(self, channels: int, *args, tail_mass: float = 1e-9, init_scale: float = 10,
                 filters: Sequence[int] = (3, 3, 3, 3), likelihood_bound: float = 1e-9,
                 entropy_coder_precision: int = 16, likelihood_form: Optional[str] = None, **kwargs):
        super().__init__()
        import os
        self.likelihood_form = likelihood_form or os.environ.get('CAE_LIKELIHOOD_FORM', 'plain')
        if self.likelihood_form not in LIKELIHOOD_FORMS:
            raise ValueError(f'likelihood_form must be one of {LIKELIHOOD_FORMS}, got {self.likelihood_form!r}')
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.entropy_coder_precision = int(entropy_coder_precision)
        self.use_likelihood_bound = likelihood_bound > 0
        if self.use_likelihood_bound:
            self.likelihood_lower_bound = LowerBound(likelihood_bound)

        self.register_buffer('_offset', torch.IntTensor())
        self.register_buffer('_quantized_cdf', torch.IntTensor())
        self.register_buffer('_cdf_length', torch.IntTensor())

        filters_ = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / filters_[i + 1]))
            matrix = torch.Tensor(channels, filters_[i + 1], filters_[i])
            matrix.data.fill_(init)
            self.register_parameter(f'_matrix{i:d}', nn.Parameter(matrix))
            bias = torch.Tensor(channels, filters_[i + 1], 1)
            nn.init.uniform_(bias, -0.5, 0.5)
            self.register_parameter(f'_bias{i:d}', nn.Parameter(bias))
            if i < len(self.filters):
                factor = torch.Tensor(channels, filters_[i + 1], 1)
                nn.init.zeros_(factor)
                self.register_parameter(f'_factor{i:d}', nn.Parameter(factor))

        self.quantiles = nn.Parameter(torch.Tensor(channels, 1, 3))
        init = torch.Tensor([-self.init_scale, 0, self.init_scale])
        self.quantiles.data = init.repeat(self.quantiles.size(0), 1, 1)
        target = np.log(2 / self.tail_mass - 1)
        self.register_buffer('target', torch.Tensor([-target, 0, target]))

        self._handle: Optional[_lib.Handle] = None
        self._tables_version = -1
        self._density_version = None

    # ---- state dict: integer buffers change size with the model --------------------------------
    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        for name in ('_offset', '_quantized_cdf', '_cdf_length'):
            key = prefix + name
            if key in state_dict:
                setattr(self, name, torch.empty_like(state_dict[key], dtype=torch.int32,
                                                     device=getattr(self, name).device))
        self._tables_version = -1
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    # ---- density model (torch ops; training / rate estimate only) ------------------------------
    def _get_medians(self) -> torch.Tensor:
        return self.quantiles[:, :, 1:2]

    def _params(self, stop_gradient: bool = False, cpu: bool = False):
        out = {}
        for k, v in self.named_parameters(recurse=False):
            if stop_gradient or cpu:
                v = v.detach()
            out[k] = v.cpu().float() if cpu else v
        return out

    def _bound_value(self) -> float:
        """the likelihood bound as a host number, read once (a device read per step would synchronise the stream)"""
        v = getattr(self, '_bound_host', None)
        if v is None:
            v = self._bound_host = float(self.likelihood_lower_bound.bound.item())
        return v

    def _fused_train(self) -> bool:
        """fused training-mode density kernels (filters all equal to the built width; CAE_EB_FUSED=0 keeps the
        element-wise graph, which the tests use as the comparison)"""
        import os
        if os.environ.get('CAE_EB_FUSED', '1') == '0' or len(set(self.filters)) != 1:
            return False
        return _lib.lib().cae_t_density_params(self.filters[0], len(self.filters)) > 0

    def _likelihood(self, inputs: torch.Tensor, stop_gradient: bool = False):
        return _likelihood(self._params(stop_gradient), len(self.filters), inputs, self.likelihood_form)

    def forward(self, x: torch.Tensor, training: Optional[bool] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        if training is None:
            training = self.training
        if not training and x.is_cuda and not torch.is_grad_enabled():
            return self._forward_hip(x)  # eval, no autograd: fused HIP kernel (cae_likelihood)
        if training and x.is_cuda and self._fused_train():
            # train mode on the GPU: one forward and one backward kernel instead of ~250 element-wise launches per step
            fixed = getattr(self, 'fixed_noise', None)
            noise = fixed if fixed is not None else torch.empty_like(x, dtype=torch.float32).uniform_(-0.5, 0.5)
            k = len(self.filters)
            params = ([getattr(self, f'_matrix{i:d}') for i in range(k + 1)] + [getattr(self, f'_bias{i:d}') for i in range(k + 1)]
                      + [getattr(self, f'_factor{i:d}') for i in range(k)])
            bound = self._bound_value() if self.use_likelihood_bound else 0.0
            return _DensityTrainFn.apply(x, noise, self.likelihood_form == 'plain', bound, *params)
        perm = list(range(x.dim()))
        perm[0], perm[1] = 1, 0
        x = x.permute(*perm).contiguous()
        shape = x.size()
        values = x.reshape(x.size(0), 1, -1)
        if training:
            # additive uniform noise (train mode, _taskutils.py:97); `fixed_noise` (same layout as the input) makes a
            # step reproducible for the parity tests
            fixed = getattr(self, 'fixed_noise', None)
            noise = (fixed.to(values).permute(*perm).reshape(values.shape) if fixed is not None
                     else torch.empty_like(values).uniform_(-0.5, 0.5))
            outputs = values + noise
        else:
            medians = self._get_medians()
            outputs = torch.round(values - medians) + medians
        likelihood, _, _ = self._likelihood(outputs)
        if self.use_likelihood_bound:
            likelihood = self.likelihood_lower_bound(likelihood)
        outputs = outputs.reshape(shape).permute(*perm).contiguous()
        likelihood = likelihood.reshape(shape).permute(*perm).contiguous()
        return outputs, likelihood

    # ---- eval-mode density on the GPU (HIP kernel) -------------------------------------------------
    def _sync_density(self) -> _lib.Handle:
        """Upload softplus(_matrix), _bias, tanh(_factor) when they changed (cae_model_set_density)."""
        if self._quantized_cdf.numel() == 0:
            self.update()
        h = self._sync_handle()
        names = [n for n, _ in self.named_parameters(recurse=False) if n != 'quantiles']
        ver = tuple((n, getattr(self, n)._version, getattr(self, n).data_ptr()) for n in names) + (self.likelihood_form,)
        if self._density_version != ver:
            k = len(self.filters)
            params = self._params(cpu=True)
            mats = [np.ascontiguousarray(F.softplus(params[f'_matrix{i:d}']).numpy(), dtype=np.float32) for i in range(k + 1)]
            bias = [np.ascontiguousarray(params[f'_bias{i:d}'].numpy(), dtype=np.float32) for i in range(k + 1)]
            fact = [np.ascontiguousarray(torch.tanh(params[f'_factor{i:d}']).numpy(), dtype=np.float32) for i in range(k)]
            filt = (ctypes.c_int * max(k, 1))(*self.filters)
            pm = (ctypes.c_void_p * (k + 1))(*[a.ctypes.data for a in mats])
            pb = (ctypes.c_void_p * (k + 1))(*[a.ctypes.data for a in bias])
            pf = (ctypes.c_void_p * (k + 1))(*([a.ctypes.data for a in fact] + [None]))
            bound = float(self.likelihood_lower_bound.bound.item()) if self.use_likelihood_bound else 0.0
            _lib.check(_lib.lib().cae_model_set_density(h.ptr, self.channels, k, filt, pm, pb, pf, bound))
            _lib.check(_lib.lib().cae_model_set_likelihood_form(h.ptr, LIKELIHOOD_FORMS.index(self.likelihood_form)))
            self._density_version = ver
        return h

    def _likelihood_hip(self, x: torch.Tensor, want_outputs: bool, want_bits: bool):
        dev = _lib.require_gpu()
        if x.dim() < 3 or x.size(1) != self.channels:
            raise ValueError(f'Invalid input shape {tuple(x.shape)} for {self.channels} channels')
        h = self._sync_density()
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        hw = int(np.prod(x.shape[2:]))
        y_hat = torch.empty_like(x) if want_outputs else None
        lik = torch.empty_like(x) if want_outputs else None
        bits = torch.empty(x.size(0), dtype=torch.float64, device=dev) if want_bits else None
        ptr = lambda t: t.data_ptr() if t is not None else None  # noqa: E731
        _lib.check(_lib.lib().cae_likelihood(h.ptr, x.data_ptr(), x.size(0), hw, ptr(y_hat), ptr(lik), ptr(bits),
                                             _lib.stream_ptr()))
        return y_hat, lik, bits

    def _forward_hip(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        y_hat, lik, _ = self._likelihood_hip(x, True, False)
        return y_hat, lik

    @torch.no_grad()
    def rate_bits(self, x: torch.Tensor) -> torch.Tensor:
        """Estimated code length per batch item, -sum log2 p(y_hat) (float64, on the GPU): the numerator of
        the reference's RateLoss (models/criteria/_ratedist.py:49-54) without materialising p."""
        return self._likelihood_hip(x, False, True)[2]

    def loss(self) -> torch.Tensor:
        logits = _logits_cumulative(self._params(stop_gradient=True), len(self.filters), self.quantiles)
        return torch.abs(logits - self.target).sum()

    @torch.no_grad()
    def fit_quantiles(self, iters: int = 60):
        """Move ``quantiles`` to the fixed point of ``loss()`` (logits(q) = target) by bisection.

        This is what the reference's ``_aux`` optimiser converges to during training
        (train_cae_ms.py:592-596); used to give synthetic, untrained models a realistic CDF support.
        """
        q = self.quantiles.detach().clone().cpu()
        params = self._params(cpu=True)

        def logits(v):
            return _logits_cumulative(params, len(self.filters), v)

        target = self.target.detach().cpu().view(1, 1, 3)
        lo = torch.full_like(q, -1e4)
        hi = torch.full_like(q, 1e4)
        for _ in range(iters):
            mid = 0.5 * (lo + hi)
            above = logits(mid) > target
            hi = torch.where(above, mid, hi)
            lo = torch.where(above, lo, mid)
        self.quantiles.data.copy_((0.5 * (lo + hi)).to(self.quantiles.device))
        self._tables_version = -1

    # ---- integer tables ----------------------------------------------------------------------
    @torch.no_grad()
    def update(self, force: bool = False) -> bool:
        if self._offset.numel() > 0 and not force:
            return False
        dev = self.quantiles.device
        params = self._params(cpu=True)  # CDFs are built on the host with torch-CPU fp32 ops
        q = params['quantiles']
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)
        samples = samples[None, :] + pmf_start[:, None, None]
        pmf, lower, upper = _likelihood(params, len(self.filters), samples, self.likelihood_form)
        pmf = pmf[:, 0, :]
        tail_mass = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        cdf = torch.zeros((len(pmf_length), max_length + 2), dtype=torch.int32)
        for i, p in enumerate(pmf):
            prob = torch.cat((p[:pmf_length[i]], tail_mass[i]), dim=0)
            c = pmf_to_quantized_cdf(prob, self.entropy_coder_precision)
            cdf[i, :c.size(0)] = c
        self._offset = offset.to(dev)
        self._quantized_cdf = cdf.to(dev)
        self._cdf_length = (pmf_length + 2).int().to(dev)
        self._tables_version = -1
        return True

    # ---- coding (hot path) ---------------------------------------------------------------------
    def _sync_handle(self) -> _lib.Handle:
        if self._quantized_cdf.numel() == 0:
            raise ValueError('Uninitialized CDFs. Run update() first')
        ver = self.tables_version()
        if self._handle is not None and self._tables_version == ver:
            return self._handle  # (the coder entry points are called from many threads at once: codec front door)
        with _SYNC_MU:
            if self._handle is None:
                self._handle = _lib.Handle(1, 1, self.channels, 1, 3)
            if self._tables_version != ver:
                self.upload_tables(self._handle)
                self._tables_version = ver
        return self._handle

    def tables_version(self):
        return (self._quantized_cdf._version, self.quantiles._version, self._quantized_cdf.data_ptr())

    def upload_tables(self, handle: _lib.Handle) -> None:
        """cae_model_set_entropy on `handle` (this module's own, or a track's for the fused quantiser)."""
        if self._quantized_cdf.numel() == 0:
            raise ValueError('Uninitialized CDFs. Run update() first')
        cdf = np.ascontiguousarray(self._quantized_cdf.detach().cpu().numpy(), dtype=np.int32)
        if cdf.ndim != 2 or cdf.shape[0] != self.channels:
            raise ValueError(f'Invalid CDF size {tuple(cdf.shape)}')
        lens = np.ascontiguousarray(self._cdf_length.detach().cpu().numpy().reshape(-1), dtype=np.int32)
        offs = np.ascontiguousarray(self._offset.detach().cpu().numpy().reshape(-1), dtype=np.int32)
        med = np.ascontiguousarray(self._get_medians().detach().cpu().numpy().reshape(-1), dtype=np.float32)
        _lib.check(_lib.lib().cae_model_set_entropy(handle.ptr, self.channels, cdf.shape[1], cdf.ctypes.data,
                                                    lens.ctypes.data, offs.ctypes.data, med.ctypes.data))

    @torch.no_grad()
    def quantize_symbols(self, x: torch.Tensor) -> torch.Tensor:
        """(B,C,...) float -> int32 symbols on the GPU (HIP quantiser kernel)."""
        dev = _lib.require_gpu()
        h = self._sync_handle()
        if x.dim() < 3 or x.size(1) != self.channels:
            raise ValueError(f'Invalid input shape {tuple(x.shape)} for {self.channels} channels')
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        hw = int(np.prod(x.shape[2:]))
        sym = torch.empty(x.shape, dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().cae_quantize(h.ptr, x.data_ptr(), x.size(0), hw, sym.data_ptr(), _lib.stream_ptr()))
        return sym

    @torch.no_grad()
    def quantize_export(self, x: torch.Tensor, out_pinned: torch.Tensor, max_blocks: int = 32) -> None:
        """Quantise (B,C,...) latents on the current stream straight into a pinned host int32 tensor of the same
        number of elements (cae_quantize_export): the symbols cross PCIe from a few workgroups beside the
        compute kernels.  Synchronise the stream before reading ``out_pinned``."""
        dev = _lib.require_gpu()
        h = self._sync_handle()
        if x.dim() < 3 or x.size(1) != self.channels:
            raise ValueError(f'Invalid input shape {tuple(x.shape)} for {self.channels} channels')
        if (not out_pinned.is_pinned() or out_pinned.dtype != torch.int32 or out_pinned.numel() != x.numel()
                or not out_pinned.is_contiguous()):
            raise ValueError('out_pinned must be a contiguous pinned int32 tensor with one element per latent')
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        hw = int(np.prod(x.shape[2:]))
        _lib.check(_lib.lib().cae_quantize_export(h.ptr, x.data_ptr(), x.size(0), hw, out_pinned.data_ptr(),
                                                  int(max_blocks), _lib.stream_ptr()))

    @torch.no_grad()
    def dequantize_symbols(self, sym: torch.Tensor) -> torch.Tensor:
        """(B,C,...) int32 symbols on the GPU -> float latents symbols + median (HIP kernel)."""
        dev = _lib.require_gpu()
        h = self._sync_handle()
        sym = sym.to(device=dev, dtype=torch.int32).contiguous()
        hw = int(np.prod(sym.shape[2:]))
        out = torch.empty(sym.shape, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().cae_dequantize(h.ptr, sym.data_ptr(), sym.size(0), hw, out.data_ptr(),
                                             _lib.stream_ptr()))
        return out

    def encode_symbols(self, sym_host: np.ndarray, threads: int = 0, packed: bool = False):
        """(B,C,hw) int32 host symbols -> one rANS byte string per batch item (``packed``: a PackedStreams, no
        per-stream copies -- 17 MB of Python-side copying per 32 tiles of 1024^2 otherwise)."""
        h = self._sync_handle()
        sym_host = np.ascontiguousarray(sym_host, dtype=np.int32)
        n = sym_host.shape[0]
        hw = int(np.prod(sym_host.shape[2:])) if sym_host.ndim > 2 else 1
        if packed:
            buf = ctypes.c_void_p()
            offs = (ctypes.c_size_t * (n + 1))()
            _lib.check(_lib.lib().cae_rans_encode_packed(h.ptr, sym_host.ctypes.data, n, hw, ctypes.byref(buf), offs,
                                                         threads))
            return PackedStreams(buf.value, list(offs))
        bufs = (ctypes.c_void_p * n)()
        lens = (ctypes.c_size_t * n)()
        _lib.check(_lib.lib().cae_rans_encode_batch(h.ptr, sym_host.ctypes.data, n, hw, bufs, lens, threads))
        out = []
        for i in range(n):
            out.append(ctypes.string_at(bufs[i], lens[i]))
            _lib.lib().cae_free(bufs[i])
        return out

    def decode_symbols(self, strings: Sequence[bytes], hw: int, threads: int = 0,
                       out: Optional[np.ndarray] = None) -> np.ndarray:
        """rANS byte strings -> (B,C,hw) int32 host symbols (written into `out` when given)."""
        h = self._sync_handle()
        n = len(strings)
        if isinstance(strings, PackedStreams):  # zero-copy: pointers into the encoder's buffer
            bufs = (ctypes.c_void_p * n)(*[strings.pointer(i) for i in range(n)])
            lens = (ctypes.c_size_t * n)(*[strings.nbytes(i) for i in range(n)])
        else:
            keep = [s if isinstance(s, bytes) else bytes(s) for s in strings]
            bufs = (ctypes.c_char_p * n)(*keep)
            lens = (ctypes.c_size_t * n)(*[len(s) for s in keep])
        if out is None:
            sym = np.empty((n, self.channels, hw), dtype=np.int32)
        else:
            sym = out
            if sym.shape != (n, self.channels, hw) or sym.dtype != np.int32 or not sym.flags.c_contiguous:
                raise ValueError('out must be a C-contiguous int32 array of shape (B,C,hw)')
        _lib.check(_lib.lib().cae_rans_decode_batch(h.ptr, bufs, lens, n, hw, sym.ctypes.data, threads))
        return sym

    @torch.no_grad()
    def compress(self, x: torch.Tensor) -> List[bytes]:
        sym = self.quantize_symbols(x)
        return self.encode_symbols(sym.reshape(sym.size(0), sym.size(1), -1).cpu().numpy())

    @torch.no_grad()
    def decompress(self, strings: Sequence[bytes], size) -> torch.Tensor:
        dev = _lib.require_gpu()
        h = self._sync_handle()
        size = tuple(int(s) for s in size)
        hw = int(np.prod(size))
        sym = torch.from_numpy(self.decode_symbols(strings, hw)).to(dev)
        out = torch.empty((len(strings), self.channels) + size, dtype=torch.float32, device=dev)
        _lib.check(_lib.lib().cae_dequantize(h.ptr, sym.data_ptr(), len(strings), hw, out.data_ptr(),
                                             _lib.stream_ptr()))
        return out
