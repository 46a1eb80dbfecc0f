"""Analysis / synthesis stacks with the reference's ``nn.Module`` call surface.

Mirrors ``src/models/tasks/_autoencoders.py`` of the reference:
``DownsamplingUnit`` :53-101, ``UpsamplingUnit`` :177-227, ``Analyzer`` :307-361,
``Synthesizer`` :364-455, ``initialize_weights`` :37-42 and ``compressai.layers.GDN``
(call site :29-30).  Same constructor arguments, same state-dict keys
(``analysis_track.{i}.model.{0|1}.{weight|bias|beta|gamma}``), same return values -- but
``forward`` runs the whole track as fused HIP kernels through libcae_hip.so.  The parameters
live in ordinary ``nn.Parameter`` tensors (so checkpoints, ``.cuda()``, optimisers and
``load_state_dict`` keep working) and are re-packed into the device-side MFMA layout whenever
they change.  There is no CPU execution path.
"""
from __future__ import annotations

import ctypes
import math
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from .entropy import LowerBound

REPARAM_OFFSET = 2.0 ** -18


class RangeTicket:
    """Deferred range check of one call on the f16x3 kernels (include/cae_hip.h, "VALID RANGE of f16x3").

    ``overflowed()`` may be asked once the stream work of the call has completed (the caller synchronised with
    it); ``rerun()`` repeats the call on the exact-fp32 kernels and returns its result."""

    def __init__(self, track, handle, ticket: int, call):
        self._track, self._handle, self.ticket, self._call = track, handle, int(ticket), call

    def overflowed(self) -> bool:
        if self.ticket == 0:
            return False  # the call already ran on the fp32 kernels
        over = ctypes.c_int(0)
        _lib.check(_lib.lib().cae_range_check(self._handle.ptr, self.ticket, ctypes.byref(over)))
        return bool(over.value)

    def rerun(self):
        self._track.fp32_fallbacks += 1
        L = _lib.lib()
        L.cae_thread_force_fp32(1)
        try:
            return self._call()
        finally:
            L.cae_thread_force_fp32(0)


class _ReparamFn(torch.autograd.Function):
    """max(x, bound)^2 - pedestal with the LowerBound gradient rule, one kernel each way (cae_t_reparam_*): as torch ops
    the twelve GDN parameters of the canonical model cost ~120 launches per training step."""

    @staticmethod
    def forward(ctx, x, bound, pedestal):
        xc = x.detach().contiguous().float()
        out = torch.empty_like(xc)
        _lib.check(_lib.lib().cae_t_reparam_forward(xc.data_ptr(), xc.numel(), bound, pedestal, out.data_ptr(),
                                                    _lib.stream_ptr()))
        ctx.save_for_backward(xc)
        ctx.bound = bound
        return out

    @staticmethod
    def backward(ctx, g):
        (xc,) = ctx.saved_tensors
        g = g.contiguous().float()
        gx = torch.empty_like(xc)
        _lib.check(_lib.lib().cae_t_reparam_backward(xc.data_ptr(), g.data_ptr(), xc.numel(), ctx.bound, gx.data_ptr(),
                                                     _lib.stream_ptr()))
        return gx, None, None


class NonNegativeParametrizer(nn.Module):
    """compressai.ops.parametrizers.NonNegativeParametrizer."""

    def __init__(self, minimum: float = 0, reparam_offset: float = REPARAM_OFFSET):
        super().__init__()
        self.minimum = float(minimum)
        self.reparam_offset = float(reparam_offset)
        pedestal = self.reparam_offset ** 2
        self.register_buffer('pedestal', torch.Tensor([pedestal]))
        bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
        self.lower_bound = LowerBound(bound)

    def init(self, x: torch.Tensor) -> torch.Tensor:
        return torch.sqrt(torch.max(x + self.pedestal, self.pedestal))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        import os
        if (x.is_cuda and x.requires_grad and torch.is_grad_enabled()
                and os.environ.get('CAE_REPARAM_FUSED', '1') != '0'):  # training on the GPU: fused kernels
            bound = (self.minimum + self.reparam_offset ** 2) ** 0.5
            return _ReparamFn.apply(x, float(np.float32(bound)), float(np.float32(self.reparam_offset ** 2)))
        out = self.lower_bound(x)
        return out ** 2 - self.pedestal


class GDN(nn.Module):
    """Parameter holder + stand-alone forward for compressai.layers.GDN."""

    def __init__(self, in_channels: int, inverse: bool = False, beta_min: float = 1e-6, gamma_init: float = 0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.in_channels = int(in_channels)
        self.beta_reparam = NonNegativeParametrizer(minimum=float(beta_min))
        self.beta = nn.Parameter(self.beta_reparam.init(torch.ones(in_channels)))
        self.gamma_reparam = NonNegativeParametrizer()
        self.gamma = nn.Parameter(self.gamma_reparam.init(float(gamma_init) * torch.eye(in_channels)))
        self._owner = None  # (track module, layer index), set by the owning Analyzer / Synthesizer

    @torch.no_grad()
    def effective(self) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.beta_reparam(self.beta), self.gamma_reparam(self.gamma)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self._owner is None:
            raise RuntimeError('GDN layer is not attached to an Analyzer/Synthesizer')
        track, index = self._owner
        return track()._gdn_forward(index, x)


class _ConvParams(nn.Module):
    """Weight/bias holder standing where the reference has nn.Conv2d / nn.ConvTranspose2d."""

    transposed = False

    def __init__(self, channels_in: int, channels_out: int, kernel_size: int, bias: bool, groups: int = 1):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size = channels_in, channels_out, kernel_size
        self.groups = int(groups)
        if channels_in % self.groups != 0:
            raise ValueError('in_channels must be divisible by groups')
        if channels_out % self.groups != 0:
            raise ValueError('out_channels must be divisible by groups')
        # parameter shapes of nn.Conv2d / nn.ConvTranspose2d with `groups` (state-dict compatible)
        shape = ((channels_in, channels_out // self.groups) if self.transposed
                 else (channels_out, channels_in // self.groups))
        self.weight = nn.Parameter(torch.empty(*shape, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(channels_out)) if bias else None
        # default nn.Conv2d init first (keeps the RNG stream aligned with the reference ctor) ...
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            fan_in = self.weight.size(1) * kernel_size * kernel_size
            bound = 1 / math.sqrt(fan_in) if fan_in > 0 else 0
            nn.init.uniform_(self.bias, -bound, bound)

    def dense_weight(self) -> torch.Tensor:
        """Weight of the equivalent groups=1 layer: (cout,cin,k,k), transposed (cin,cout,k,k).  A grouped layer is
        the dense one with a block-diagonal channel matrix; the dense MFMA kernels then compute it exactly (the
        zero products add 0.0), at the dense layer's cost."""
        w = self.weight.detach().float()
        if self.groups == 1:
            return w
        g, k = self.groups, self.kernel_size
        cin_g, cout_g = self.in_channels // g, self.out_channels // g
        if self.transposed:
            dense = w.new_zeros(self.in_channels, self.out_channels, k, k)
            for j in range(g):
                dense[j * cin_g:(j + 1) * cin_g, j * cout_g:(j + 1) * cout_g] = w[j * cin_g:(j + 1) * cin_g]
        else:
            dense = w.new_zeros(self.out_channels, self.in_channels, k, k)
            for j in range(g):
                dense[j * cout_g:(j + 1) * cout_g, j * cin_g:(j + 1) * cin_g] = w[j * cout_g:(j + 1) * cout_g]
        return dense

    def forward(self, x):
        raise NotImplementedError('layers run fused: call the owning Analyzer / Synthesizer')


class StridedReflectConv2d(_ConvParams):
    transposed = False


class StridedConvTranspose2d(_ConvParams):
    transposed = True


def initialize_weights(m):
    """reference initialize_weights (_autoencoders.py:37-42)."""
    if isinstance(m, _ConvParams):
        nn.init.xavier_uniform_(m.weight.data, gain=math.sqrt(2 / 1.01))
        if m.bias is not None:
            nn.init.constant_(m.bias.data, 0.01)


_ACT_CODES = {None: 0, 'Identity': 0, 'GDN': 0, 'LeakyReLU': 1, 'ReLU': 2}


def _define_act_layer(act_layer_type, channels_in=None, track='analysis'):
    """reference _define_act_layer (_autoencoders.py:19-34)."""
    if act_layer_type is None or act_layer_type == 'Identity':
        return nn.Identity()
    if act_layer_type == 'LeakyReLU':
        return nn.LeakyReLU(inplace=False)
    if act_layer_type == 'ReLU':
        return nn.ReLU(inplace=False)
    if act_layer_type == 'GDN':
        return GDN(in_channels=channels_in, inverse=track == 'synthesis')
    raise ValueError(f'Activation layer {act_layer_type} not supported')


def _check_variant(kernel_size, groups, batch_norm, dropout, use_residual, act_layer_type, channels_expansion):
    if act_layer_type not in (None, 'GDN', 'LeakyReLU', 'ReLU'):
        raise ValueError(f'Activation layer {act_layer_type} not supported')
    if kernel_size not in (3, 5):
        raise NotImplementedError('kernel_size must be 3 or 5')
    if int(channels_expansion) < 1:
        raise ValueError('channels_expansion must be >= 1')
    del groups, batch_norm, dropout, use_residual  # grouped / batch-normalised layers are folded on upload; Dropout2d is the identity in eval


def _fold_batch_norm(w: torch.Tensor, b: Optional[torch.Tensor], bn: Optional[nn.BatchNorm2d], transposed: bool):
    """Eval-mode BatchNorm2d behind a convolution = the same convolution with scaled weights and a shifted bias
    (running statistics; the codec path is inference only)."""
    if bn is None:
        return w, None if b is None else b.detach().float()
    if bn.training:
        raise NotImplementedError('BatchNorm2d in training mode (batch statistics) is not built: call .eval()')
    scale = (bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps))
    w = w * (scale.view(1, -1, 1, 1) if transposed else scale.view(-1, 1, 1, 1))
    b0 = torch.zeros_like(scale) if b is None else b.detach().float()
    return w, (b0 - bn.running_mean.detach().float()) * scale + bn.bias.detach().float()


class _Unit(nn.Module):
    """model = [pre conv (stride 1) + act]? + strided (transposed) conv + [act | GDN]?  -- the module order,
    hence the state-dict indices, of the reference's DownsamplingUnit / UpsamplingUnit."""

    _conv_cls = StridedReflectConv2d
    _track = 'analysis'

    def __init__(self, channels_in, channels_out, kernel_size=3, groups=False, batch_norm=False, dropout=0.0,
                 bias=False, act_layer_type=None):
        super().__init__()
        model = []
        self.pre_index = self.gdn_index = self.pre_bn_index = self.main_bn_index = None
        g = channels_in if groups else 1
        if act_layer_type is not None and act_layer_type not in ['GDN']:
            self.pre_index = 0
            model.append(self._conv_cls(channels_in, channels_in, kernel_size, bias, g))
            if batch_norm:
                self.pre_bn_index = len(model)
                model.append(nn.BatchNorm2d(channels_in, affine=True))
            model.append(_define_act_layer(act_layer_type, channels_in, track=self._track))
        self.main_index = len(model)
        model.append(self._conv_cls(channels_in, channels_out, kernel_size, bias, g))
        if batch_norm:
            self.main_bn_index = len(model)
            model.append(nn.BatchNorm2d(channels_out, affine=True))
        if act_layer_type is not None:
            if act_layer_type == 'GDN':
                self.gdn_index = len(model)
            model.append(_define_act_layer(act_layer_type, channels_out, track=self._track))
        if dropout > 0.0:
            model.append(nn.Dropout2d(dropout))
        self.act_code = _ACT_CODES[act_layer_type]
        self.model = nn.Sequential(*model)

    @property
    def main(self):
        return self.model[self.main_index]

    @property
    def pre(self):
        return None if self.pre_index is None else self.model[self.pre_index]

    @property
    def gdn(self):
        return None if self.gdn_index is None else self.model[self.gdn_index]

    def effective_main(self):
        """(dense weight, bias | None) of the strided layer with groups expanded and BatchNorm folded."""
        conv = self.main
        bn = None if self.main_bn_index is None else self.model[self.main_bn_index]
        return _fold_batch_norm(conv.dense_weight(), conv.bias, bn, conv.transposed)

    def effective_pre(self):
        conv = self.pre
        bn = None if self.pre_bn_index is None else self.model[self.pre_bn_index]
        return _fold_batch_norm(conv.dense_weight(), conv.bias, bn, conv.transposed)


class DownsamplingUnit(_Unit):
    pass


class UpsamplingUnit(_Unit):
    _conv_cls = StridedConvTranspose2d
    _track = 'synthesis'

    def __init__(self, channels_in, channels_out, kernel_size=3, groups=False, batch_norm=False, dropout=0.0,
                 bias=True, act_layer_type=None):
        super().__init__(channels_in, channels_out, kernel_size, groups, batch_norm, dropout, bias, act_layer_type)


class _ResidualUnit(nn.Module):
    """res_model (stride-1 convolutions on the unit input) + unit input, then model (the strided layer): module
    order, hence state-dict indices, of the reference's ResidualDownsamplingUnit / ResidualUpsamplingUnit
    (_autoencoders.py:104-174, :230-304).  On the device the stride-1 convolutions are `stages` of the layer
    (cae_model_set_layer_stage): y = post_act(act_or_gdn(conv(x)) [+ unit input])."""

    _conv_cls = StridedReflectConv2d
    _track = 'analysis'
    _second_stage_act = False  # the synthesis unit activates its second stride-1 convolution, the analysis one not

    def __init__(self, channels_in, channels_out, kernel_size=3, groups=False, batch_norm=False, dropout=0.0,
                 bias=False, act_layer_type=None):
        super().__init__()
        g = channels_in if groups else 1
        plain_act = act_layer_type is not None and act_layer_type not in ['GDN']
        res_model = [self._conv_cls(channels_in, channels_in, kernel_size, bias, g)]
        self._res = [[0, None, None]]  # per stage: [conv index, bn index, gdn index] in res_model
        if batch_norm:
            self._res[0][1] = len(res_model)
            res_model.append(nn.BatchNorm2d(channels_in, affine=True))
        if act_layer_type == 'GDN':
            self._res[0][2] = len(res_model)
        res_model.append(_define_act_layer(act_layer_type, channels_in, track=self._track))
        if plain_act:
            self._res.append([len(res_model), None, None])
            res_model.append(self._conv_cls(channels_in, channels_in, kernel_size, bias, g))
            if batch_norm:
                self._res[1][1] = len(res_model)
                res_model.append(nn.BatchNorm2d(channels_in, affine=True))
            if self._second_stage_act:
                res_model.append(_define_act_layer(act_layer_type, channels_in, track=self._track))
        model = []
        if plain_act:
            # (the reference sizes this activation with channels_out; irrelevant for LeakyReLU / ReLU, Appendix B)
            model.append(_define_act_layer(act_layer_type, channels_out, track=self._track))
        self.main_index = len(model)
        model.append(self._conv_cls(channels_in, channels_out, kernel_size, bias, g))
        self.main_bn_index = self.gdn_index = None
        if batch_norm:
            self.main_bn_index = len(model)
            model.append(nn.BatchNorm2d(channels_out, affine=True))
        if act_layer_type is not None:
            if act_layer_type == 'GDN':
                self.gdn_index = len(model)
            model.append(_define_act_layer(act_layer_type, channels_out, track=self._track))
        if dropout > 0.0:
            model.append(nn.Dropout2d(dropout))
        self.act_code = _ACT_CODES[act_layer_type]
        self.res_model = nn.Sequential(*res_model)
        self.model = nn.Sequential(*model)

    pre = None  # (the stride-1 convolutions are described by stages())

    @property
    def main(self):
        return self.model[self.main_index]

    @property
    def gdn(self):
        return None if self.gdn_index is None else self.model[self.gdn_index]

    def effective_main(self):
        conv = self.main
        bn = None if self.main_bn_index is None else self.model[self.main_bn_index]
        return _fold_batch_norm(conv.dense_weight(), conv.bias, bn, conv.transposed)

    def stages(self):
        """[dict(weight, bias, beta, gamma, act, add_residual, post_act)] for cae_model_set_layer_stage."""
        out = []
        last = len(self._res) - 1
        for k, (ci, bi, gi) in enumerate(self._res):
            conv = self.res_model[ci]
            w, b = _fold_batch_norm(conv.dense_weight(), conv.bias, None if bi is None else self.res_model[bi],
                                    conv.transposed)
            beta = gamma = None
            if gi is not None:
                beta, gamma = self.res_model[gi].effective()
            act = self.act_code if (k == 0 or self._second_stage_act) else 0
            out.append(dict(weight=w, bias=b, beta=beta, gamma=gamma, act=0 if gi is not None else act,
                            add_residual=int(k == last), post_act=self.act_code if (k == last and last == 1) else 0))
        return out


class ResidualDownsamplingUnit(_ResidualUnit):
    pass


class ResidualUpsamplingUnit(_ResidualUnit):
    _conv_cls = StridedConvTranspose2d
    _track = 'synthesis'
    _second_stage_act = True

    def __init__(self, channels_in, channels_out, kernel_size=3, groups=False, batch_norm=False, dropout=0.0,
                 bias=True, act_layer_type=None):
        super().__init__(channels_in, channels_out, kernel_size, groups, batch_norm, dropout, bias, act_layer_type)


class _Track(nn.Module):
    """Shared device plumbing of Analyzer / Synthesizer."""

    _track_id = _lib.CAE_ANALYSIS
    _track_attr = 'analysis_track'

    def _setup(self, channels_org, channels_net, channels_bn, compression_level, kernel_size):
        self._dims = (int(channels_org), int(channels_net), int(channels_bn), int(compression_level),
                      int(kernel_size))
        self._handle: Optional[_lib.Handle] = None
        self._versions = None
        self.fp32_fallbacks = 0  # calls repeated on the fp32 kernels because a value left the f16 range
        import weakref
        ref = weakref.ref(self)
        for i, unit in enumerate(getattr(self, self._track_attr)):
            if unit.gdn is not None:
                unit.gdn._owner = (ref, i)

    def precision_code(self) -> int:
        """0 = exact fp32 MFMA, 1 = f16x3 split MFMA.  Attribute `precision` ('fp32' | 'f16x3'), else the
        CAE_PRECISION environment variable, else 'f16x3' (fp32-class accuracy at ~2x the throughput;
        GDN layers wider than 128 channels run their normalisation as a separate f16x3 kernel).  Colour layers to more
        than 32 image channels are built on the fp32 kernels only."""
        import os
        prec = getattr(self, 'precision', None) or os.environ.get('CAE_PRECISION', 'f16x3')
        if prec not in ('fp32', 'f16x3'):
            raise ValueError(f"precision must be 'fp32' or 'f16x3', got {prec!r}")
        # (residual units: the split-f16 stride-1 kernel carries the (I)GDN / residual-sum epilogue up to 128 channels; the
        #  library takes the stages of a wider unit through its fp32 kernels, the strided layers stay on f16x3)
        # multiscale colour layers: split-f16 stride-1 kernel to at most 32 image channels
        if getattr(self, 'multiscale_analysis', False) and self._dims[0] > 32:
            return 0
        # (LeakyReLU / ReLU units: the stride-1 pre-convolutions run on the split-f16 kernel too, conv_s2_f16_kernel<.., S = 1>;
        #  the 192-channel transposed-convolution kernel carries no activation epilogue -- it would spill)
        if self._track_id == _lib.CAE_SYNTHESIS and any(u.act_code and u.main.out_channels > 128 for u in self._units()):
            return 0
        return 1 if prec == 'f16x3' else 0

    def _units(self):
        return list(getattr(self, self._track_attr))

    def _param_versions(self):
        # buffers too (BatchNorm running statistics), the mode (BatchNorm is folded in eval mode only) and the
        # arithmetic path: changing `precision` / CAE_PRECISION re-packs the layers for the other kernels
        return (tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))
                + (self.training, self.precision_code()))

    def _guarded(self, hd, call, defer: bool = False):
        """Runs ``call()`` (allocates the outputs, launches, returns them) under the f16x3 range guard: if a value
        left the f16 range the call is repeated on the fp32 kernels.  ``defer``: no synchronisation here; returns
        (outputs, RangeTicket) and the caller checks once it has synchronised with the call anyway."""
        out = call()
        ticket = RangeTicket(self, hd, _lib.lib().cae_last_range_ticket(), call)
        if defer:
            return out, ticket
        if ticket.ticket:
            torch.cuda.current_stream().synchronize()
            if ticket.overflowed():
                out = ticket.rerun()
        return out

    def _sync(self) -> _lib.Handle:
        _lib.require_gpu()
        if self._handle is None:
            self._handle = _lib.Handle(*self._dims)
        ver = self._param_versions()
        if ver != self._versions:
            L = _lib.lib()
            _lib.check(L.cae_model_set_precision(self._handle.ptr, ver[-1]))
            with torch.no_grad():
                for i, unit in enumerate(self._units()):
                    conv = unit.main
                    wt, bt = unit.effective_main()
                    w = np.ascontiguousarray(wt.cpu().numpy(), dtype=np.float32)
                    b = None if bt is None else np.ascontiguousarray(bt.detach().float().cpu().numpy())
                    beta = gamma = None
                    if unit.gdn is not None:
                        be, ga = unit.gdn.effective()
                        beta = np.ascontiguousarray(be.float().cpu().numpy())
                        gamma = np.ascontiguousarray(ga.float().cpu().numpy())
                    _lib.check(L.cae_model_set_layer(
                        self._handle.ptr, self._track_id, i, conv.in_channels, conv.out_channels,
                        w.ctypes.data, None if b is None else b.ctypes.data,
                        None if beta is None else beta.ctypes.data, None if gamma is None else gamma.ctypes.data))
                    if isinstance(unit, _ResidualUnit):
                        _lib.check(L.cae_model_set_layer_act(self._handle.ptr, self._track_id, i, unit.act_code, None,
                                                             None))
                        for k, sg in enumerate(unit.stages()):
                            arr = {key: (None if sg[key] is None else
                                         np.ascontiguousarray(sg[key].detach().float().cpu().numpy()))
                                   for key in ('weight', 'bias', 'beta', 'gamma')}
                            ptr = {key: (None if v is None else v.ctypes.data) for key, v in arr.items()}
                            _lib.check(L.cae_model_set_layer_stage(
                                self._handle.ptr, self._track_id, i, k, ptr['weight'], ptr['bias'], ptr['beta'],
                                ptr['gamma'], sg['act'], sg['add_residual'], sg['post_act']))
                    elif unit.act_code or unit.pre is not None:
                        pw = pb = None
                        if unit.pre is not None:
                            pwt, pbt = unit.effective_pre()
                            pw = np.ascontiguousarray(pwt.cpu().numpy(), dtype=np.float32)
                            if pbt is not None:
                                pb = np.ascontiguousarray(pbt.detach().float().cpu().numpy())
                        _lib.check(L.cae_model_set_layer_act(
                            self._handle.ptr, self._track_id, i, unit.act_code,
                            None if pw is None else pw.ctypes.data, None if pb is None else pb.ctypes.data))
            if getattr(self, 'multiscale_analysis', False):
                with torch.no_grad():
                    for i, layer in enumerate(list(self.color_layers)[:-1]):
                        conv = layer[0]
                        cw = np.ascontiguousarray(conv.dense_weight().cpu().numpy(), dtype=np.float32)
                        cb = None if conv.bias is None else np.ascontiguousarray(conv.bias.detach().float().cpu().numpy())
                        _lib.check(L.cae_model_set_color_layer(self._handle.ptr, i, conv.in_channels,
                                                               conv.out_channels, cw.ctypes.data,
                                                               None if cb is None else cb.ctypes.data))
            self._versions = ver
        return self._handle

    def _sync_entropy(self, eb) -> _lib.Handle:
        """This track's handle with `eb`'s medians loaded (fused quantiser / dequantiser entry points)."""
        hd = self._sync()
        ver = (id(eb), eb.tables_version())
        if getattr(self, '_entropy_version', None) != ver:
            eb.upload_tables(hd)
            self._entropy_version = ver
        return hd

    def set_profiling(self, enable: bool):
        """Bracket every kernel of this track with HIP events (see cae_model_set_profiling)."""
        _lib.check(_lib.lib().cae_model_set_profiling(self._sync().ptr, int(bool(enable))))

    def get_profile(self, reset: bool = True):
        """-> (ms per slot summed over calls [layout conversion, layer 0, layer 1, ...], number of calls)."""
        n = self._dims[3] + 1
        ms = (ctypes.c_double * n)()
        calls = ctypes.c_int(0)
        _lib.check(_lib.lib().cae_model_get_profile(self._sync().ptr, self._track_id, ms, n, ctypes.byref(calls),
                                                    int(reset)))
        return list(ms), calls.value

    def _gdn_forward(self, index: int, x: torch.Tensor) -> torch.Tensor:
        dev = _lib.require_gpu()
        h = self._sync()
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        if x.dim() != 4:
            raise ValueError(f'expected a 4-D (B,C,H,W) tensor, got {tuple(x.shape)}')
        y = torch.empty_like(x)
        _lib.check(_lib.lib().cae_gdn_forward(h.ptr, self._track_id, index, x.data_ptr(), x.size(0), x.size(2),
                                              x.size(3), y.data_ptr(), _lib.stream_ptr()))
        return y


class Analyzer(_Track):
    def __init__(self, channels_org=3, channels_net=8, channels_bn=16, compression_level=3, channels_expansion=1,
                 kernel_size=3, groups=False, batch_norm=False, dropout=0.0, bias=False, use_residual=False,
                 act_layer_type=None, **kwargs):
        super().__init__()
        _check_variant(kernel_size, groups, batch_norm, dropout, use_residual, act_layer_type, channels_expansion)
        if compression_level < 1:
            raise NotImplementedError('compression_level must be >= 1')
        down_track = []
        DownsamplingUnit_ = ResidualDownsamplingUnit if use_residual else DownsamplingUnit
        prev, curr = channels_org, channels_net
        for _ in range(compression_level - 1):
            down_track.append(DownsamplingUnit_(prev, curr, kernel_size, groups, batch_norm, dropout, bias,
                                                act_layer_type))
            prev = curr
            curr = prev * channels_expansion  # reference channel plan (_autoencoders.py:330-342)
        down_track.append(DownsamplingUnit_(prev, channels_bn, kernel_size, groups, batch_norm, dropout, bias, None))
        self.analysis_track = nn.Sequential(*down_track)
        self.apply(initialize_weights)
        self._setup(channels_org, channels_net, channels_bn, compression_level, kernel_size)

    def latent_size(self, h: int, w: int) -> Tuple[int, int]:
        for _ in range(self._dims[3]):
            h, w = (h + 1) // 2, (w + 1) // 2
        return h, w

    def _run(self, ptr: int, fmt: int, n: int, h: int, w: int) -> torch.Tensor:
        dev = _lib.require_gpu()
        hd = self._sync()
        lh, lw = self.latent_size(h, w)

        def call():
            y = torch.empty((n, self._dims[2], lh, lw), dtype=torch.float32, device=dev)
            _lib.check(_lib.lib().cae_analysis(hd.ptr, ptr, fmt, n, h, w, y.data_ptr(), _lib.stream_ptr()))
            return y
        return self._guarded(hd, call)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x (B,C,H,W) float in [0,1] -> y (B,channels_bn,ceil(H/2^L),ceil(W/2^L))."""
        dev = _lib.require_gpu()
        if x.dim() != 4 or x.size(1) != self._dims[0]:
            raise ValueError(f'expected (B,{self._dims[0]},H,W), got {tuple(x.shape)}')
        from . import train
        if train.needs_grad(self, x):  # autograd is recording: bf16 / fp32-GDN training kernels with HIP backward
            return train.analysis_forward(self, x)
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        return self._run(x.data_ptr(), _lib.FMT_F32_NCHW, x.size(0), x.size(2), x.size(3))

    def forward_u8(self, tiles: torch.Tensor) -> torch.Tensor:
        """tiles (B,H,W,C) uint8 on the GPU -> latents; fuses the /255 of codec.encode."""
        dev = _lib.require_gpu()
        if tiles.dim() != 4 or tiles.size(3) != self._dims[0] or tiles.dtype != torch.uint8:
            raise ValueError(f'expected uint8 (B,H,W,{self._dims[0]}), got {tiles.dtype} {tuple(tiles.shape)}')
        tiles = tiles.to(dev).contiguous()
        return self._run(tiles.data_ptr(), _lib.FMT_U8_HWC, tiles.size(0), tiles.size(1), tiles.size(2))


    def forward_u8_symbols(self, tiles: torch.Tensor, eb, defer: bool = False):
        """tiles (B,H,W,C) uint8 on the GPU -> int32 symbols round(y - median) of `eb` (an EntropyBottleneck):
        analysis with the quantiser fused into the last layer's epilogue (cae_analysis_symbols); equals
        eb.quantize_symbols(self.forward_u8(tiles)).  ``defer``: -> (symbols, RangeTicket), see _guarded."""
        dev = _lib.require_gpu()
        if tiles.dim() != 4 or tiles.size(3) != self._dims[0] or tiles.dtype != torch.uint8:
            raise ValueError(f'expected uint8 (B,H,W,{self._dims[0]}), got {tiles.dtype} {tuple(tiles.shape)}')
        if eb.channels != self._dims[2]:
            raise ValueError(f'entropy model has {eb.channels} channels, the latents {self._dims[2]}')
        tiles = tiles.to(dev).contiguous()
        hd = self._sync_entropy(eb)
        n, h, w = tiles.size(0), tiles.size(1), tiles.size(2)
        lh, lw = self.latent_size(h, w)

        def call():
            sym = torch.empty((n, self._dims[2], lh, lw), dtype=torch.int32, device=dev)
            _lib.check(_lib.lib().cae_analysis_symbols(hd.ptr, tiles.data_ptr(), _lib.FMT_U8_HWC, n, h, w,
                                                       sym.data_ptr(), _lib.stream_ptr()))
            return sym
        return self._guarded(hd, call, defer)


class NoneColorLayer(nn.Module):
    def forward(self, *args, **kwargs):
        return None


class Synthesizer(_Track):
    _track_id = _lib.CAE_SYNTHESIS
    _track_attr = 'synthesis_track'

    def __init__(self, channels_org=3, channels_net=8, channels_bn=16, compression_level=3, channels_expansion=1,
                 kernel_size=3, groups=False, batch_norm=False, dropout=0.0, bias=False, use_residual=False,
                 act_layer_type=None, multiscale_analysis=False, **kwargs):
        super().__init__()
        _check_variant(kernel_size, groups, batch_norm, dropout, use_residual, act_layer_type, channels_expansion)
        self.multiscale_analysis = bool(multiscale_analysis)
        if compression_level < 1:
            raise NotImplementedError('compression_level must be >= 1')
        up_track = []
        UpsamplingUnit_ = ResidualUpsamplingUnit if use_residual else UpsamplingUnit
        # reference channel plan (_autoencoders.py:385-400): starts at net * e^L and divides by e per level
        prev, curr = channels_bn, channels_net * channels_expansion ** compression_level
        for _ in range(compression_level - 1):
            up_track.append(UpsamplingUnit_(prev, curr, kernel_size, groups, batch_norm, dropout, bias, act_layer_type))
            prev = curr
            curr = prev // channels_expansion
        up_track.append(UpsamplingUnit_(prev, channels_org, kernel_size, groups, batch_norm, dropout, bias, None))
        self.synthesis_track = nn.Sequential(*up_track)
        if multiscale_analysis:
            # reference channel plan (_autoencoders.py:417-428): net * e^i, i descending; it matches the track's
            # channels only for channels_expansion == 1 (checked when the weights are uploaded)
            color_layers = [nn.Sequential(StridedReflectConv2d(channels_net * channels_expansion ** i, channels_org,
                                                               kernel_size, bias, channels_org if groups else 1))
                            for i in reversed(range(compression_level - 1))]
        else:
            color_layers = [nn.Sequential(NoneColorLayer()) for _ in range(compression_level - 1)]
        color_layers += [nn.Identity()]
        self.color_layers = nn.ModuleList(color_layers)
        self.rec_level = compression_level
        self.apply(initialize_weights)
        self._setup(channels_org, channels_net, channels_bn, compression_level, kernel_size)

    def _run(self, x: torch.Tensor, fmt: int, bridges: bool, colors: bool = False):
        dev = _lib.require_gpu()
        hd = self._sync()
        if x.dim() != 4 or x.size(1) != self._dims[2]:
            raise ValueError(f'expected (B,{self._dims[2]},h,w), got {tuple(x.shape)}')
        x = x.detach().to(device=dev, dtype=torch.float32).contiguous()
        n, _, lh, lw = x.shape
        L = self._dims[3]
        H, W = lh * 2 ** L, lw * 2 ** L

        def call():
            if fmt == _lib.FMT_U8_HWC:
                out = torch.empty((n, H, W, self._dims[0]), dtype=torch.uint8, device=dev)
            else:
                out = torch.empty((n, self._dims[0], H, W), dtype=torch.float32, device=dev)
            brg: List[torch.Tensor] = []
            brg_ptr = None
            if bridges and L > 1:
                brg = [torch.empty((n, u.main.out_channels, lh * 2 ** (i + 1), lw * 2 ** (i + 1)),
                                   dtype=torch.float32, device=dev) for i, u in enumerate(self._units()[:-1])]
                brg_ptr = (ctypes.c_void_p * (L - 1))(*[b.data_ptr() for b in brg])
            col: List[torch.Tensor] = []
            col_ptr = None
            if colors and self.multiscale_analysis and L > 1:
                col = [torch.empty((n, self._dims[0], lh * 2 ** (i + 1), lw * 2 ** (i + 1)), dtype=torch.float32,
                                   device=dev) for i in range(L - 1)]
                col_ptr = (ctypes.c_void_p * (L - 1))(*[c.data_ptr() for c in col])
            _lib.check(_lib.lib().cae_synthesis_multiscale(hd.ptr, x.data_ptr(), n, lh, lw, out.data_ptr(), fmt,
                                                           brg_ptr, col_ptr, _lib.stream_ptr()))
            return out, brg, col
        return self._guarded(hd, call)

    def forward(self, x: torch.Tensor, bridges: bool = True):
        """y_q (B,channels_bn,h,w) -> (x_r list [full-res, half-res | None, ...], fx_brg list) as the reference:
        x_r[0] is the reconstruction, x_r[j] the colour layer of level L-1-j (None without multiscale_analysis)."""
        from . import train
        if train.needs_grad(self, x):  # autograd is recording (training): see train.synthesis_forward
            if x.dim() != 4 or x.size(1) != self._dims[2]:
                raise ValueError(f'expected (B,{self._dims[2]},h,w), got {tuple(x.shape)}')
            return train.synthesis_forward(self, x)
        out, brg, col = self._run(x, _lib.FMT_F32_NCHW, bridges, colors=True)
        x_r = [out] + (col[::-1] if col else [None] * (self._dims[3] - 1))
        return x_r, brg + [out]

    def forward_u8(self, x: torch.Tensor) -> torch.Tensor:
        """y_q -> (B,H,W,C) uint8 tiles; fuses the *255 / clip / truncate / HWC epilogue of codec.decode."""
        out, _, _ = self._run(x, _lib.FMT_U8_HWC, False)
        return out

    def forward_symbols_u8(self, sym: torch.Tensor, eb, defer: bool = False):
        """int32 symbols (B,channels_bn,h,w) on the GPU -> (B,H,W,C) uint8 tiles: the dequantiser of `eb` fused into
        the layout conversion in front of the first layer (cae_synthesis_symbols); equals
        self.forward_u8(eb.dequantize_symbols(sym)).  ``defer``: -> (tiles, RangeTicket), see _guarded."""
        dev = _lib.require_gpu()
        if sym.dim() != 4 or sym.size(1) != self._dims[2] or sym.dtype != torch.int32:
            raise ValueError(f'expected int32 (B,{self._dims[2]},h,w), got {sym.dtype} {tuple(sym.shape)}')
        if eb.channels != self._dims[2]:
            raise ValueError(f'entropy model has {eb.channels} channels, the latents {self._dims[2]}')
        sym = sym.to(dev).contiguous()
        hd = self._sync_entropy(eb)
        n, _, lh, lw = sym.shape
        L = self._dims[3]

        def call():
            out = torch.empty((n, lh * 2 ** L, lw * 2 ** L, self._dims[0]), dtype=torch.uint8, device=dev)
            _lib.check(_lib.lib().cae_synthesis_symbols(hd.ptr, sym.data_ptr(), n, lh, lw, out.data_ptr(),
                                                        _lib.FMT_U8_HWC, _lib.stream_ptr()))
            return out
        return self._guarded(hd, call, defer)
