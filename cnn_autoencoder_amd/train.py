"""Training path on the HIP kernels (BASELINE config 5).

What the reference's training step does (``src/train_cae_ms.py:189-262``) and where it lives here:

* ``forward_func`` (``models/tasks/_taskutils.py:95-108``): encoder -> fact_ent (train mode: additive U(-1/2, 1/2)
  noise) -> decoder.  ``Analyzer.forward`` / ``Synthesizer.forward`` switch to the differentiable track functions of
  this module whenever autograd is recording: bf16 convolutions with fp32 accumulation
  (``cae_t_conv_forward`` / ``cae_t_deconv_forward``), fp32 GDN / IGDN (``cae_t_gdn_forward``), and hand-written
  backward kernels (data gradients, weight gradients, GDN gradient) instead of ATen / cuDNN autograd.
* ``GeneralLoss`` (``models/criteria/_lossutils.py:54-109``): ``criteria.GeneralLoss`` (scalar reductions, torch ops).
* compressai's ``NonNegativeParametrizer`` / ``LowerBound`` gradient rule of the GDN parameters: the kernels
  differentiate with respect to the EFFECTIVE beta / gamma, the reparametrisation stays a torch autograd graph
  (``modules.NonNegativeParametrizer``), so the rule (pass where ``p >= bound`` or ``grad < 0``) is applied exactly
  once, on parameter-sized tensors.
* ``setup_optim`` (``train_cae_ms.py:529-655``): one optimiser per trainable module, the ``quantiles`` of the entropy
  model in a separate ``<module>_aux`` optimiser (:592-596); ``train_step`` = :209-230 (loss.backward, aux_loss.backward,
  per-optimiser clip_grad_norm_(1.0), step, zero_grad).
* ``nn.DataParallel``'s gradient reduction (``_autoencoders.py:517``): ``GradReducer``, bucketed all-reduce over
  ``torch.distributed`` (RCCL over xGMI on GPUs, gloo on CPU), one process per GPU.

Variants covered: ``act_layer_type in (None, 'GDN', 'LeakyReLU', 'ReLU')`` units (the last two with their stride-1
pre-convolutions) on the fused track functions, and residual units (``use_residual=True``) composed per operation from
the same kernels (``_composed_track``), as are units with batch norm in training mode (batch statistics, running
statistics updated as ``nn.BatchNorm2d`` does), grouped layers (dense kernels on the block-diagonal embedding of the
grouped weight) and ``Dropout2d``; multiscale colour layers raise ``NotImplementedError`` under autograd.
"""
from __future__ import annotations

import ctypes
from typing import Dict, List, Optional, Sequence, Tuple

import os

import torch
import torch.nn as nn

from . import _lib


def _pad32(c: int) -> int:
    return (int(c) + 31) // 32 * 32


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _L():
    return _lib.lib()


def _st():
    return _lib.stream_ptr()


class LayerSpec:
    """Static description of one unit for the track functions."""

    def __init__(self, cin: int, cout: int, ks: int, has_bias: bool, has_gdn: bool, act: int = 0, has_pre: bool = False):
        self.cin, self.cout, self.ks = int(cin), int(cout), int(ks)
        self.cin_p, self.cout_p = _pad32(cin), _pad32(cout)
        self.has_bias, self.has_gdn = bool(has_bias), bool(has_gdn)
        # LeakyReLU (1) / ReLU (2) units: activation after the strided layer, and (has_pre) a stride-1 convolution
        # cin -> cin + the same activation in front of it (_autoencoders.py:62-76, :187-202)
        self.act, self.has_pre = int(act), bool(has_pre)

    @property
    def n_tensors(self) -> int:
        return (1 + int(self.has_bias)) * (1 + int(self.has_pre)) + 2 * int(self.has_gdn)


def _pack(weight: torch.Tensor, contract_dim: int, ks: int) -> torch.Tensor:
    """fp32 (d0, d1, k, k) -> bf16 MFMA B fragments on the device (cae_t_pack_weights)."""
    d0, d1 = weight.shape[0], weight.shape[1]
    kc, nc = (d0, d1) if contract_dim == 0 else (d1, d0)
    nbytes = _L().cae_t_packed_bytes(kc, nc, ks)
    out = torch.empty(nbytes // 2, dtype=torch.bfloat16, device=weight.device)
    w = weight.detach().float().contiguous()
    _lib.check(_L().cae_t_pack_weights(w.data_ptr(), d0, d1, ks, contract_dim, out.data_ptr(), _st()))
    return out


def _split_params(specs: Sequence[LayerSpec], tensors: Sequence[torch.Tensor]):
    """-> per layer (w, b, beta, gamma, pre_w, pre_b); flat order: [pre_w, pre_b?]? w, b?, [beta, gamma]?"""
    out, k = [], 0
    for s in specs:
        pw = pb = None
        if s.has_pre:
            pw = tensors[k]
            k += 1
            if s.has_bias:
                pb = tensors[k]
                k += 1
        w = tensors[k]
        k += 1
        b = beta = gamma = None
        if s.has_bias:
            b = tensors[k]
            k += 1
        if s.has_gdn:
            beta, gamma = tensors[k], tensors[k + 1]
            k += 2
        out.append((w, b, beta, gamma, pw, pb))
    return out


def _bias_p(b: Optional[torch.Tensor], cp: int, dev) -> Optional[torch.Tensor]:
    if b is None:
        return None
    out = torch.zeros(cp, dtype=torch.float32, device=dev)
    out[:b.numel()] = b.detach().float()
    return out


def _edge_ok(spec: 'LayerSpec', c_img: int) -> bool:
    """the 3-channel edge of a track as a pointwise GEMM over K = (tap, channel) <= 32 (cae_t_im2col_s2 / cae_t_col2im_s2):
    no stride-1 pre-convolution at that edge, k^2 * channels fits one 32-deep chunk; CAE_EDGE_GEMM=0 keeps the padded form"""
    return (not spec.has_pre and spec.ks * spec.ks * c_img <= 32 and os.environ.get('CAE_EDGE_GEMM', '1') != '0')


def _act_backward(g16, gext32, pad, y16, act):
    """gradient through LeakyReLU / ReLU: g * (y > 0 ? 1 : slope), y = the activation's output; g bf16, or the fp32
    extended-domain gradient (folded in place first) -> bf16"""
    n, h, w, cp = y16.shape
    out = torch.empty_like(y16)
    _lib.check(_L().cae_t_act_backward(_ptr(g16), _ptr(gext32), pad, y16.data_ptr(), n, h, w, cp, act, out.data_ptr(), _st()))
    return out


def _colsum(g16, c):
    n, h, w, cp = g16.shape
    gb = torch.empty(cp, dtype=torch.float32, device=g16.device)
    _lib.check(_L().cae_t_colsum(g16.data_ptr(), n * h * w, cp, gb.data_ptr(), _st()))
    return gb[:c].clone()


def _flat_grads(specs, per_layer):
    """per_layer[i] = [g_w, g_b, g_beta, g_gamma, g_pre_w, g_pre_b] -> the flat order of _split_params"""
    out: List[Optional[torch.Tensor]] = []
    for s, (g_w, g_b, g_beta, g_gamma, g_pw, g_pb) in zip(specs, per_layer):
        if s.has_pre:
            out.append(g_pw)
            if s.has_bias:
                out.append(g_pb)
        out.append(g_w)
        if s.has_bias:
            out.append(g_b)
        if s.has_gdn:
            out.extend([g_beta, g_gamma])
    return out


def _from_nchw(x: torch.Tensor, cp: int, want16=True, want32=False):
    n, c, h, w = x.shape
    x = x.detach().float().contiguous()
    o16 = torch.empty((n, h, w, cp), dtype=torch.bfloat16, device=x.device) if want16 else None
    o32 = torch.empty((n, h, w, cp), dtype=torch.float32, device=x.device) if want32 else None
    _lib.check(_L().cae_t_from_nchw(x.data_ptr(), n, c, h, w, cp, _ptr(o16), _ptr(o32), _st()))
    return o16, o32


def _to_nchw(t32: torch.Tensor, c: int) -> torch.Tensor:
    n, h, w, cp = t32.shape
    out = torch.empty((n, c, h, w), dtype=torch.float32, device=t32.device)
    _lib.check(_L().cae_t_to_nchw(t32.data_ptr(), n, c, h, w, cp, out.data_ptr(), _st()))
    return out


def _gdn_fused(cp: int) -> bool:
    """one-kernel GDN forward (saving the factor f, y = z f) / backward (cae_t_gdn_*_save / _fused: up to 128 channels);
    CAE_GDN_FUSED=0 keeps the three-kernel backward, which the tests use as the comparison"""
    return cp <= 128 and os.environ.get('CAE_GDN_FUSED', '1') != '0'


def _gdn_forward(z32: torch.Tensor, beta_p: torch.Tensor, gamma_p: torch.Tensor, inverse: bool):
    """-> (y16, saved factor f | None)"""
    n, h, w, cp = z32.shape
    y16 = torch.empty_like(z32, dtype=torch.bfloat16)
    beta, gamma = beta_p.detach().float().contiguous(), gamma_p.detach().float().contiguous()  # (alive across the call)
    if _gdn_fused(cp):
        f = torch.empty(_L().cae_t_gdn_saved_elems(n * h * w, cp), dtype=torch.float32, device=z32.device)
        _lib.check(_L().cae_t_gdn_forward_save(z32.data_ptr(), n * h * w, cp, beta.data_ptr(), gamma.data_ptr(),
                                               int(inverse), y16.data_ptr(), f.data_ptr(), _st()))
        return y16, f
    _lib.check(_L().cae_t_gdn_forward(z32.data_ptr(), n * h * w, cp, beta.data_ptr(), gamma.data_ptr(), int(inverse),
                                      None, y16.data_ptr(), _st()))
    return y16, None


def _gdn_backward(z32, gext32, pad, beta_p, gamma_p, inverse, f=None):
    """-> (gz16, g_beta_p, g_gamma_p)"""
    n, h, w, cp = z32.shape
    dev = z32.device
    if f is not None:
        gz16 = torch.empty_like(z32, dtype=torch.bfloat16)
        gg = torch.empty((cp, cp), dtype=torch.float32, device=dev)
        gb = torch.empty((cp,), dtype=torch.float32, device=dev)
        gamma = gamma_p.detach().float().contiguous()
        _lib.check(_L().cae_t_gdn_backward_fused(z32.data_ptr(), f.data_ptr(), gext32.data_ptr(), n, h, w, pad, cp,
                                                 gamma.data_ptr(), int(inverse), gz16.data_ptr(), gg.data_ptr(),
                                                 gb.data_ptr(), _st()))
        return gz16, gb, gg
    gn = torch.empty_like(z32)
    gzd = torch.empty_like(z32)
    gz16 = torch.empty_like(z32, dtype=torch.bfloat16)
    gg = torch.empty((cp, cp), dtype=torch.float32, device=dev)
    gb = torch.empty((cp,), dtype=torch.float32, device=dev)
    gamma = gamma_p.detach().float().contiguous()
    gamma_t = gamma.t().contiguous()
    beta = beta_p.detach().float().contiguous()
    _lib.check(_L().cae_t_gdn_backward(z32.data_ptr(), gext32.data_ptr(), n, h, w, pad, cp,
                                       beta.data_ptr(), gamma.data_ptr(), gamma_t.data_ptr(),
                                       int(inverse), gn.data_ptr(), gzd.data_ptr(), None, gz16.data_ptr(), gg.data_ptr(),
                                       gb.data_ptr(), _st()))
    return gz16, gb, gg


def _weight_grad(gw: torch.Tensor, spec_shape: Tuple[int, int], ks: int) -> torch.Tensor:
    """gw [k*k][ca][cb] -> gradient of a (d0 = b, d1 = a, k, k) weight"""
    d0, d1 = spec_shape
    return gw.permute(2, 1, 0)[:d0, :d1].reshape(d0, d1, ks, ks).contiguous()


class AnalysisFn(torch.autograd.Function):
    """Analyzer.forward under autograd: L x [(conv s1 + act)? reflect conv s2 (+bias) (+GDN | act)]
    (_autoencoders.py:62-85, :29-30)."""

    @staticmethod
    def forward(ctx, x, specs, *tensors):
        L = _L()
        layers = _split_params(specs, tensors)
        n, _, h, w = x.shape
        dev = x.device
        edge = _edge_ok(specs[0], specs[0].cin)
        a16 = None if edge else _from_nchw(x, specs[0].cin_p)[0]
        saved, dims = [], []
        z32 = None
        for i, (s, (wt, b, beta, gamma, pw, pb)) in enumerate(zip(specs, layers)):
            last = i == len(specs) - 1
            oh, ow = (h + 1) // 2, (w + 1) // 2
            if i == 0 and edge:
                # first layer: K = (tap, channel) = 27 of 32 as ONE contraction chunk of a 1 x 1 GEMM on the im2col of the image
                cols = torch.empty((n, oh, ow, 32), dtype=torch.bfloat16, device=dev)
                xc = x.detach().float().contiguous()
                _lib.check(L.cae_t_im2col_s2(xc.data_ptr(), n, s.cin, h, w, oh, ow, s.ks, 1, cols.data_ptr(), _st()))
                w1 = torch.zeros((s.cout, 32, 1, 1), dtype=torch.float32, device=dev)
                w1[:, :s.ks * s.ks * s.cin, 0, 0] = wt.detach().float().permute(0, 2, 3, 1).reshape(s.cout, -1)
                wp = _pack(w1, 1, 1)
                bias_p = _bias_p(b, s.cout_p, dev)
                need32 = s.has_gdn or last
                z32 = torch.empty((n, oh, ow, s.cout_p), dtype=torch.float32, device=dev) if need32 else None
                z16 = None if need32 else torch.empty((n, oh, ow, s.cout_p), dtype=torch.bfloat16, device=dev)
                _lib.check(L.cae_t_pointwise(cols.data_ptr(), n, oh, ow, 32, wp.data_ptr(), _ptr(z32), _ptr(z16), s.cout_p,
                                             _ptr(bias_p), 0 if need32 else s.act, _st()))
                f_saved = None
                if s.has_gdn:
                    a16, f_saved = _gdn_forward(z32, beta, gamma, False)
                elif not last:
                    a16 = z16
                saved.append(dict(a_in=cols, p=None, z=z32 if s.has_gdn else None, f=f_saved,
                                  out=a16 if (s.act and not need32) else None, edge=True))
                dims.append((h, w, oh, ow))
                h, w = oh, ow
                continue
            a16_in, p16 = a16, None
            if s.has_pre:  # stride-1 reflect convolution cin -> cin + activation
                p16 = torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
                wpp, bpp = _pack(pw, 1, s.ks), _bias_p(pb, s.cin_p, dev)
                _lib.check(L.cae_t_corr_s1(a16.data_ptr(), n, h, w, s.cin_p, wpp.data_ptr(), s.ks, 0, None, p16.data_ptr(),
                                           s.cin_p, _ptr(bpp), s.act, _st()))
            wp = _pack(wt, 1, s.ks)
            bias_p = _bias_p(b, s.cout_p, dev)
            need32 = s.has_gdn or last
            z32 = torch.empty((n, oh, ow, s.cout_p), dtype=torch.float32, device=dev) if need32 else None
            z16 = None if need32 else torch.empty((n, oh, ow, s.cout_p), dtype=torch.bfloat16, device=dev)
            main_in = p16 if p16 is not None else a16
            _lib.check(L.cae_t_conv_forward_act(main_in.data_ptr(), n, h, w, s.cin_p, wp.data_ptr(), s.ks, _ptr(z32), _ptr(z16),
                                                s.cout_p, _ptr(bias_p), 0 if need32 else s.act, _st()))
            f_saved = None
            if s.has_gdn:
                a16, f_saved = _gdn_forward(z32, beta, gamma, False)
            elif not last:
                a16 = z16  # (post-activation output of a LeakyReLU / ReLU unit)
            saved.append(dict(a_in=a16_in, p=p16, z=z32 if s.has_gdn else None, f=f_saved,
                              out=a16 if (s.act and not need32) else None))
            dims.append((h, w, oh, ow))
            h, w = oh, ow
        y = _to_nchw(z32, specs[-1].cout)
        ctx.specs, ctx.saved, ctx.dims, ctx.layers = specs, saved, dims, [tuple(t.detach() if t is not None else None
                                                                                 for t in l) for l in layers]
        return y

    @staticmethod
    def backward(ctx, gy):
        L = _L()
        specs, saved, dims, layers = ctx.specs, ctx.saved, ctx.dims, ctx.layers
        n = gy.shape[0]
        dev = gy.device
        g16, _ = _from_nchw(gy, specs[-1].cout_p)  # gradient with respect to the last convolution's output
        per_layer = [[None] * 6 for _ in specs]
        masked = False  # g16 already went through this unit's activation (applied on the fp32 gradient: one rounding)
        for i in reversed(range(len(specs))):
            s = specs[i]
            wt, b, _, _, pw, pb = layers[i]
            sv = saved[i]
            h, w, oh, ow = dims[i]
            kk = s.ks * s.ks
            P = s.ks // 2
            if sv['out'] is not None and not masked:  # through the unit's activation
                g16 = _act_backward(g16, None, 0, sv['out'], s.act)
            masked = False
            if sv.get('edge'):  # first layer as a pointwise GEMM: its weight gradient over the im2col, back in (cout, cin, k, k)
                gw1 = torch.empty((1, 32, s.cout_p), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_wgrad_pointwise(sv['a_in'].data_ptr(), g16.data_ptr(), n, oh, ow, 32, s.cout_p,
                                                   gw1.data_ptr(), _st()))
                per_layer[i][0] = (gw1[0, :kk * s.cin, :s.cout].t().reshape(s.cout, s.ks, s.ks, s.cin)
                                   .permute(0, 3, 1, 2).contiguous())
                if b is not None:
                    per_layer[i][1] = _colsum(g16, s.cout)
                break  # (the image itself needs no gradient)
            main_in = sv['p'] if s.has_pre else sv['a_in']
            gw = torch.empty((kk, s.cin_p, s.cout_p), dtype=torch.float32, device=dev)
            _lib.check(L.cae_t_wgrad(main_in.data_ptr(), n, h, w, s.cin_p, g16.data_ptr(), oh, ow, s.cout_p, s.ks, 1,
                                     gw.data_ptr(), _st()))
            per_layer[i][0] = _weight_grad(gw, (s.cout, s.cin), s.ks)
            if b is not None:
                per_layer[i][1] = _colsum(g16, s.cout)
            if i == 0 and not s.has_pre:
                break  # (the image itself needs no gradient)
            wp_d = _pack(wt, 0, s.ks)
            gext = torch.empty((n, h + 2 * P, w + 2 * P, s.cin_p), dtype=torch.float32, device=dev)
            _lib.check(L.cae_t_conv_dgrad_ext(g16.data_ptr(), n, oh, ow, s.cout_p, wp_d.data_ptr(), s.ks, h, w,
                                              gext.data_ptr(), s.cin_p, _st()))
            if s.has_pre:
                gu16 = _act_backward(None, gext, P, sv['p'], s.act)  # fold + activation mask: gradient at the pre-convolution's output
                gwp = torch.empty((kk, s.cin_p, s.cin_p), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_wgrad_s1(sv['a_in'].data_ptr(), n, h, w, s.cin_p, gu16.data_ptr(), s.cin_p, s.ks, 1,
                                            gwp.data_ptr(), _st()))
                per_layer[i][4] = _weight_grad(gwp, (s.cin, s.cin), s.ks)
                if pb is not None:
                    per_layer[i][5] = _colsum(gu16, s.cin)
                if i == 0:
                    break
                wpp_d = _pack(pw, 0, s.ks)
                gext = torch.empty((n, h + 2 * P, w + 2 * P, s.cin_p), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_corr_s1(gu16.data_ptr(), n, h, w, s.cin_p, wpp_d.data_ptr(), s.ks, 1, gext.data_ptr(), None,
                                           s.cin_p, None, 0, _st()))
            if specs[i - 1].has_gdn:
                z_prev, f_prev = saved[i - 1]['z'], saved[i - 1]['f']
                _, _, beta_p, gamma_p, _, _ = layers[i - 1]
                g16, g_beta, g_gamma = _gdn_backward(z_prev, gext, P, beta_p, gamma_p, False, f_prev)
                per_layer[i - 1][2], per_layer[i - 1][3] = g_beta, g_gamma
            elif saved[i - 1]['out'] is not None:  # reflect fold + the previous unit's activation on the fp32 gradient
                g16 = _act_backward(None, gext, P, saved[i - 1]['out'], specs[i - 1].act)
                masked = True
            else:
                g16 = torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
                _lib.check(L.cae_t_fold_to_bf16(gext.data_ptr(), n, h, w, P, s.cin_p, g16.data_ptr(), _st()))
        return (None, None, *_flat_grads(specs, per_layer))


class SynthesisFn(torch.autograd.Function):
    """Synthesizer.forward under autograd: L x [(conv-transpose s1 + act)? conv-transpose s2 (+bias) (+IGDN | act)]
    (_autoencoders.py:187-211)."""

    @staticmethod
    def forward(ctx, yq, specs, *tensors):
        L = _L()
        layers = _split_params(specs, tensors)
        n, _, h, w = yq.shape
        dev = yq.device
        a16, _ = _from_nchw(yq, specs[0].cin_p)
        saved, dims = [], []
        z32 = None
        for i, (s, (wt, b, beta, gamma, pw, pb)) in enumerate(zip(specs, layers)):
            last = i == len(specs) - 1
            a16_in, p16 = a16, None
            if s.has_pre:  # ConvTranspose2d(cin, cin, k, stride 1, padding k//2) + activation
                p16 = torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
                wpp, bpp = _pack(pw, 0, s.ks), _bias_p(pb, s.cin_p, dev)
                _lib.check(L.cae_t_corr_s1(a16.data_ptr(), n, h, w, s.cin_p, wpp.data_ptr(), s.ks, 2, None, p16.data_ptr(),
                                           s.cin_p, _ptr(bpp), s.act, _st()))
            if last and _edge_ok(s, s.cout):
                # last layer: per INPUT position the k*k*cout <= 32 products with the weights (a 1 x 1 GEMM), then col2im
                K = s.ks * s.ks * s.cout
                w1 = torch.zeros((32, s.cin, 1, 1), dtype=torch.float32, device=dev)  # (j = tap * cout + co, ci)
                w1[:K, :, 0, 0] = wt.detach().float().permute(2, 3, 1, 0).reshape(K, s.cin)
                wp = _pack(w1, 1, 1)
                main_in = p16 if p16 is not None else a16
                u32 = torch.empty((n, h, w, 32), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_pointwise(main_in.data_ptr(), n, h, w, s.cin_p, wp.data_ptr(), u32.data_ptr(), None, 32,
                                             None, 0, _st()))
                x_r = torch.empty((n, s.cout, 2 * h, 2 * w), dtype=torch.float32, device=dev)
                bias_c = None if b is None else b.detach().float().contiguous()
                _lib.check(L.cae_t_col2im_s2(u32.data_ptr(), _ptr(bias_c), n, s.cout, h, w, s.ks, x_r.data_ptr(), _st()))
                saved.append(dict(a_in=a16_in, p=p16, z=None, f=None, out=None, edge=True))
                dims.append((h, w))
                ctx.specs, ctx.saved, ctx.dims = specs, saved, dims
                ctx.layers = [tuple(t.detach() if t is not None else None for t in l) for l in layers]
                ctx.need_input_grad = yq.requires_grad
                return x_r
            wp = _pack(wt, 0, s.ks)
            bias_p = _bias_p(b, s.cout_p, dev)
            need32 = s.has_gdn or last
            z32 = torch.empty((n, 2 * h, 2 * w, s.cout_p), dtype=torch.float32, device=dev) if need32 else None
            z16 = None if need32 else torch.empty((n, 2 * h, 2 * w, s.cout_p), dtype=torch.bfloat16, device=dev)
            main_in = p16 if p16 is not None else a16
            _lib.check(L.cae_t_deconv_forward_act(main_in.data_ptr(), n, h, w, s.cin_p, wp.data_ptr(), s.ks, _ptr(z32),
                                                  _ptr(z16), s.cout_p, _ptr(bias_p), 0 if need32 else s.act, _st()))
            f_saved = None
            if s.has_gdn:
                a16, f_saved = _gdn_forward(z32, beta, gamma, True)
            elif not last:
                a16 = z16
            saved.append(dict(a_in=a16_in, p=p16, z=z32 if s.has_gdn else None, f=f_saved,
                              out=a16 if (s.act and not need32) else None))
            dims.append((h, w))
            h, w = 2 * h, 2 * w
        x_r = _to_nchw(z32, specs[-1].cout)
        ctx.specs, ctx.saved, ctx.dims = specs, saved, dims
        ctx.layers = [tuple(t.detach() if t is not None else None for t in l) for l in layers]
        ctx.need_input_grad = yq.requires_grad
        return x_r

    @staticmethod
    def backward(ctx, gx):
        L = _L()
        specs, saved, dims, layers = ctx.specs, ctx.saved, ctx.dims, ctx.layers
        n = gx.shape[0]
        dev = gx.device
        edge = bool(saved[-1].get('edge'))
        g16 = None if edge else _from_nchw(gx, specs[-1].cout_p)[0]  # gradient with respect to the last layer's output
        per_layer = [[None] * 6 for _ in specs]
        g_in = None
        masked = False  # g16 already went through this unit's activation (applied on the fp32 gradient: one rounding)
        for i in reversed(range(len(specs))):
            s = specs[i]
            wt, b, _, _, pw, pb = layers[i]
            sv = saved[i]
            h, w = dims[i]
            kk = s.ks * s.ks
            if sv.get('edge'):
                # last layer as a pointwise GEMM: gu[pos][(tap, co)] = g_x[2 pos - P + tap][co] (im2col of the output
                # gradient, zeros outside); weight gradient and data gradient are 1 x 1 contractions with it
                K = kk * s.cout
                gxc = gx.detach().float().contiguous()
                gu16 = torch.empty((n, h, w, 32), dtype=torch.bfloat16, device=dev)
                _lib.check(L.cae_t_im2col_s2(gxc.data_ptr(), n, s.cout, 2 * h, 2 * w, h, w, s.ks, 0, gu16.data_ptr(), _st()))
                main_in = sv['p'] if s.has_pre else sv['a_in']
                gw1 = torch.empty((1, s.cin_p, 32), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_wgrad_pointwise(main_in.data_ptr(), gu16.data_ptr(), n, h, w, s.cin_p, 32, gw1.data_ptr(),
                                                   _st()))
                per_layer[i][0] = (gw1[0, :s.cin, :K].reshape(s.cin, s.ks, s.ks, s.cout).permute(0, 3, 1, 2).contiguous())
                if b is not None:
                    per_layer[i][1] = gxc.bfloat16().float().sum(dim=(0, 2, 3))
                if i == 0 and not ctx.need_input_grad:
                    break
                w1d = torch.zeros((s.cin, 32, 1, 1), dtype=torch.float32, device=dev)  # (ci, j)
                w1d[:, :K, 0, 0] = wt.detach().float().permute(0, 2, 3, 1).reshape(s.cin, K)
                wp_d = _pack(w1d, 1, 1)
                prev_gdn = i > 0 and specs[i - 1].has_gdn
                prev_act = i > 0 and saved[i - 1]['out'] is not None
                want32 = prev_gdn or prev_act or i == 0
                gx32 = torch.empty((n, h, w, s.cin_p), dtype=torch.float32, device=dev) if want32 else None
                gx16 = None if want32 else torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
                _lib.check(L.cae_t_pointwise(gu16.data_ptr(), n, h, w, 32, wp_d.data_ptr(), _ptr(gx32), _ptr(gx16), s.cin_p,
                                             None, 0, _st()))
                if i == 0:
                    g_in = _to_nchw(gx32, s.cin)
                elif prev_gdn:
                    z_prev, f_prev = saved[i - 1]['z'], saved[i - 1]['f']
                    _, _, beta_p, gamma_p, _, _ = layers[i - 1]
                    g16, g_beta, g_gamma = _gdn_backward(z_prev, gx32, 0, beta_p, gamma_p, True, f_prev)
                    per_layer[i - 1][2], per_layer[i - 1][3] = g_beta, g_gamma
                elif prev_act:
                    g16 = _act_backward(None, gx32, 0, saved[i - 1]['out'], specs[i - 1].act)
                    masked = True
                else:
                    g16 = gx16
                continue
            if sv['out'] is not None and not masked:  # through the unit's activation
                g16 = _act_backward(g16, None, 0, sv['out'], s.act)
            masked = False
            main_in = sv['p'] if s.has_pre else sv['a_in']
            gw = torch.empty((kk, s.cout_p, s.cin_p), dtype=torch.float32, device=dev)
            _lib.check(L.cae_t_wgrad(g16.data_ptr(), n, 2 * h, 2 * w, s.cout_p, main_in.data_ptr(), h, w, s.cin_p, s.ks, 0,
                                     gw.data_ptr(), _st()))
            per_layer[i][0] = _weight_grad(gw, (s.cin, s.cout), s.ks)
            if b is not None:
                per_layer[i][1] = _colsum(g16, s.cout)
            if i == 0 and not ctx.need_input_grad and not s.has_pre:
                break
            wp_d = _pack(wt, 1, s.ks)
            prev_gdn = i > 0 and specs[i - 1].has_gdn
            prev_act = i > 0 and saved[i - 1]['out'] is not None
            want32 = prev_gdn or prev_act or i == 0 or s.has_pre  # (fp32 into an activation's backward: one rounding)
            gx32 = torch.empty((n, h, w, s.cin_p), dtype=torch.float32, device=dev) if want32 else None
            gx16 = None if want32 else torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
            _lib.check(L.cae_t_deconv_dgrad(g16.data_ptr(), n, h, w, s.cout_p, wp_d.data_ptr(), s.ks, _ptr(gx32), _ptr(gx16),
                                            s.cin_p, _st()))
            if s.has_pre:
                gu16 = _act_backward(None, gx32, 0, sv['p'], s.act)  # gradient at the pre-convolution's output
                gwp = torch.empty((kk, s.cin_p, s.cin_p), dtype=torch.float32, device=dev)
                _lib.check(L.cae_t_wgrad_s1(gu16.data_ptr(), n, h, w, s.cin_p, sv['a_in'].data_ptr(), s.cin_p, s.ks, 0,
                                            gwp.data_ptr(), _st()))
                per_layer[i][4] = _weight_grad(gwp, (s.cin, s.cin), s.ks)
                if pb is not None:
                    per_layer[i][5] = _colsum(gu16, s.cin)
                if i == 0 and not ctx.need_input_grad:
                    break
                wpp_d = _pack(pw, 1, s.ks)
                want32 = prev_gdn or prev_act or i == 0
                gx32 = torch.empty((n, h, w, s.cin_p), dtype=torch.float32, device=dev) if want32 else None
                gx16 = None if want32 else torch.empty((n, h, w, s.cin_p), dtype=torch.bfloat16, device=dev)
                _lib.check(L.cae_t_corr_s1(gu16.data_ptr(), n, h, w, s.cin_p, wpp_d.data_ptr(), s.ks, 3, _ptr(gx32), _ptr(gx16),
                                           s.cin_p, None, 0, _st()))
            if i == 0:
                g_in = _to_nchw(gx32, s.cin)
            elif prev_gdn:
                z_prev, f_prev = saved[i - 1]['z'], saved[i - 1]['f']
                _, _, beta_p, gamma_p, _, _ = layers[i - 1]
                g16, g_beta, g_gamma = _gdn_backward(z_prev, gx32, 0, beta_p, gamma_p, True, f_prev)
                per_layer[i - 1][2], per_layer[i - 1][3] = g_beta, g_gamma
            elif prev_act:  # the previous unit's activation on the fp32 gradient
                g16 = _act_backward(None, gx32, 0, saved[i - 1]['out'], specs[i - 1].act)
                masked = True
            else:
                g16 = gx16
        return (g_in, None, *_flat_grads(specs, per_layer))


# ---- residual units (ResidualDownsamplingUnit / ResidualUpsamplingUnit, _autoencoders.py:104-174, :230-304) ----------
# and units with batch norm: their tracks are composed per operation, on NCHW fp32 tensors between operations: every
# convolution, GDN and batch norm is one of the kernels behind a small autograd function of its own, the residual sum and
# stand-alone LeakyReLU / ReLU modules are element-wise torch operations on the unit's tensors.  (The canonical tracks keep their
# fused functions above; this form pays two layout conversions per operation.)

class _ConvS1Fn(torch.autograd.Function):
    """Stride-1 convolution cin -> cin of a residual unit's res_model, (+ bias) (+ LeakyReLU / ReLU): analysis = Conv2d with
    reflect padding, synthesis = ConvTranspose2d(stride 1, padding k//2) (cae_t_corr_s1 modes 0 / 2; backward: cae_t_wgrad_s1
    and the data gradients of modes 1 / 3)."""

    @staticmethod
    def forward(ctx, x, synthesis, ks, act, w, b):
        L = _L()
        n, c, h, wd = x.shape
        cp, dev = _pad32(c), x.device
        a16, _ = _from_nchw(x, cp)
        wp = _pack(w, 0 if synthesis else 1, ks)
        bp = _bias_p(b, cp, dev)
        out32 = torch.empty((n, h, wd, cp), dtype=torch.float32, device=dev)
        out16 = torch.empty((n, h, wd, cp), dtype=torch.bfloat16, device=dev) if act else None
        _lib.check(L.cae_t_corr_s1(a16.data_ptr(), n, h, wd, cp, wp.data_ptr(), ks, 2 if synthesis else 0, out32.data_ptr(),
                                   _ptr(out16), cp, _ptr(bp), act, _st()))
        ctx.a16, ctx.out16, ctx.w = a16, out16, w.detach()
        ctx.cfg = (bool(synthesis), int(ks), int(act), b is not None, c)
        return _to_nchw(out32, c)

    @staticmethod
    def backward(ctx, g):
        L = _L()
        synthesis, ks, act, has_bias, c = ctx.cfg
        a16 = ctx.a16
        n, h, wd, cp = a16.shape
        dev = g.device
        P = ks // 2
        if act:  # through the activation on the fp32 gradient (one rounding)
            g32 = _from_nchw(g, cp, want16=False, want32=True)[1]
            gu16 = _act_backward(None, g32, 0, ctx.out16, act)
        else:
            gu16 = _from_nchw(g, cp)[0]
        gwp = torch.empty((ks * ks, cp, cp), dtype=torch.float32, device=dev)
        if synthesis:
            _lib.check(L.cae_t_wgrad_s1(gu16.data_ptr(), n, h, wd, cp, a16.data_ptr(), cp, ks, 0, gwp.data_ptr(), _st()))
        else:
            _lib.check(L.cae_t_wgrad_s1(a16.data_ptr(), n, h, wd, cp, gu16.data_ptr(), cp, ks, 1, gwp.data_ptr(), _st()))
        g_w = _weight_grad(gwp, (c, c), ks)
        g_b = _colsum(gu16, c) if has_bias else None
        if synthesis:
            wd_p = _pack(ctx.w, 1, ks)
            gx32 = torch.empty((n, h, wd, cp), dtype=torch.float32, device=dev)
            _lib.check(L.cae_t_corr_s1(gu16.data_ptr(), n, h, wd, cp, wd_p.data_ptr(), ks, 3, gx32.data_ptr(), None, cp, None, 0,
                                       _st()))
            gx = _to_nchw(gx32, c)
        else:  # extended domain, then the reflect fold
            wd_p = _pack(ctx.w, 0, ks)
            gext = torch.empty((n, h + 2 * P, wd + 2 * P, cp), dtype=torch.float32, device=dev)
            _lib.check(L.cae_t_corr_s1(gu16.data_ptr(), n, h, wd, cp, wd_p.data_ptr(), ks, 1, gext.data_ptr(), None, cp, None, 0,
                                       _st()))
            gx16 = torch.empty((n, h, wd, cp), dtype=torch.bfloat16, device=dev)
            _lib.check(L.cae_t_fold_to_bf16(gext.data_ptr(), n, h, wd, P, cp, gx16.data_ptr(), _st()))
            gx = _to_nchw(gx16.float(), c)
        return gx, None, None, None, g_w, g_b


class _ConvS2Fn(torch.autograd.Function):
    """The strided reflect convolution of an analysis unit alone, with its input gradient (AnalysisFn's first layer reads the
    image and returns none)."""

    @staticmethod
    def forward(ctx, x, ks, w, b):
        L = _L()
        n, c, h, wd = x.shape
        cout = w.shape[0]
        cin_p, cout_p, dev = _pad32(c), _pad32(cout), x.device
        oh, ow = (h + 1) // 2, (wd + 1) // 2
        a16, _ = _from_nchw(x, cin_p)
        wp = _pack(w, 1, ks)
        bp = _bias_p(b, cout_p, dev)
        z32 = torch.empty((n, oh, ow, cout_p), dtype=torch.float32, device=dev)
        _lib.check(L.cae_t_conv_forward_act(a16.data_ptr(), n, h, wd, cin_p, wp.data_ptr(), ks, z32.data_ptr(), None, cout_p,
                                            _ptr(bp), 0, _st()))
        ctx.a16, ctx.w = a16, w.detach()
        ctx.cfg = (int(ks), b is not None, c, cout, oh, ow)
        return _to_nchw(z32, cout)

    @staticmethod
    def backward(ctx, g):
        L = _L()
        ks, has_bias, c, cout, oh, ow = ctx.cfg
        a16 = ctx.a16
        n, h, wd, cin_p = a16.shape
        cout_p, dev, P, kk = _pad32(cout), g.device, ks // 2, ks * ks
        g16 = _from_nchw(g, cout_p)[0]
        gw = torch.empty((kk, cin_p, cout_p), dtype=torch.float32, device=dev)
        _lib.check(L.cae_t_wgrad(a16.data_ptr(), n, h, wd, cin_p, g16.data_ptr(), oh, ow, cout_p, ks, 1, gw.data_ptr(), _st()))
        g_w = _weight_grad(gw, (cout, c), ks)
        g_b = _colsum(g16, cout) if has_bias else None
        wp_d = _pack(ctx.w, 0, ks)
        gext = torch.empty((n, h + 2 * P, wd + 2 * P, cin_p), dtype=torch.float32, device=dev)
        _lib.check(L.cae_t_conv_dgrad_ext(g16.data_ptr(), n, oh, ow, cout_p, wp_d.data_ptr(), ks, h, wd, gext.data_ptr(), cin_p,
                                          _st()))
        gx16 = torch.empty((n, h, wd, cin_p), dtype=torch.bfloat16, device=dev)
        _lib.check(L.cae_t_fold_to_bf16(gext.data_ptr(), n, h, wd, P, cin_p, gx16.data_ptr(), _st()))
        return _to_nchw(gx16.float(), c), None, g_w, g_b


class _GdnFn(torch.autograd.Function):
    """GDN / IGDN alone on an NCHW fp32 tensor (effective, padded beta / gamma): the fused forward / backward kernels."""

    @staticmethod
    def forward(ctx, x, inverse, beta_p, gamma_p):
        c, cp = x.shape[1], beta_p.numel()
        z32 = _from_nchw(x, cp, want16=False, want32=True)[1]
        y16, f = _gdn_forward(z32, beta_p, gamma_p, bool(inverse))
        ctx.z32, ctx.f, ctx.beta_p, ctx.gamma_p, ctx.cfg = z32, f, beta_p.detach(), gamma_p.detach(), (bool(inverse), c)
        return _to_nchw(y16.float(), c)

    @staticmethod
    def backward(ctx, g):
        inverse, c = ctx.cfg
        cp = ctx.beta_p.numel()
        g32 = _from_nchw(g, cp, want16=False, want32=True)[1]  # (the backward kernel works in this buffer)
        gz16, g_beta, g_gamma = _gdn_backward(ctx.z32, g32, 0, ctx.beta_p, ctx.gamma_p, inverse, ctx.f)
        return _to_nchw(gz16.float(), c), None, g_beta, g_gamma


def _gdn_params(g, c: int):
    """effective, padded (beta, gamma) of a GDN module, in the autograd graph (LowerBound gradient rule inside)"""
    cp = _pad32(c)
    beta, gamma = g.beta_reparam(g.beta), g.gamma_reparam(g.gamma)
    beta_p = torch.cat([beta, beta.new_ones(cp - c)]) if cp > c else beta
    gamma_p = torch.nn.functional.pad(gamma, (0, cp - c, 0, cp - c)) if cp > c else gamma
    return beta_p, gamma_p


def _torch_act(t: torch.Tensor, act: int) -> torch.Tensor:
    return t if not act else (torch.nn.functional.leaky_relu(t, 0.01) if act == 1 else torch.relu(t))


class _BatchNormFn(torch.autograd.Function):
    """nn.BatchNorm2d in training mode on an NCHW fp32 tensor: batch statistics (biased variance for the normalisation, as
    torch) by cae_t_bn_moments, y = x w rstd + (b - mean w rstd) by cae_t_bn_affine; the backward is the same pair of kernels
    with dy (the per-channel coefficients are C-element float64 arithmetic).  -> (y, batch mean, biased batch variance)"""

    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        L = _L()
        x = x.detach().float().contiguous()
        n, c, h, w = x.shape
        dev = x.device
        s1 = torch.empty(c, dtype=torch.float64, device=dev)
        s2 = torch.empty(c, dtype=torch.float64, device=dev)
        _lib.check(L.cae_t_bn_moments(x.data_ptr(), x.data_ptr(), n, c, h * w, s1.data_ptr(), s2.data_ptr(), _st()))
        m = float(n * h * w)
        mean = s1 / m
        var = (s2 / m - mean * mean).clamp_min(0.0)
        rstd = torch.rsqrt(var + eps)
        wt = weight.detach().double() if weight is not None else torch.ones(c, dtype=torch.float64, device=dev)
        bs = bias.detach().double() if bias is not None else torch.zeros(c, dtype=torch.float64, device=dev)
        A = (wt * rstd).float().contiguous()
        C = (bs - mean * wt * rstd).float().contiguous()
        y = torch.empty_like(x)
        _lib.check(L.cae_t_bn_affine(x.data_ptr(), None, n, c, h * w, A.data_ptr(), None, C.data_ptr(), y.data_ptr(), _st()))
        ctx.x, ctx.stats, ctx.has = x, (mean, rstd, wt, m), (weight is not None, bias is not None)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _gm, _gv):
        L = _L()
        x = ctx.x
        mean, rstd, wt, m = ctx.stats
        n, c, h, w = x.shape
        dev = x.device
        dy = dy.detach().float().contiguous()
        s1 = torch.empty(c, dtype=torch.float64, device=dev)
        s2 = torch.empty(c, dtype=torch.float64, device=dev)
        _lib.check(L.cae_t_bn_moments(dy.data_ptr(), x.data_ptr(), n, c, h * w, s1.data_ptr(), s2.data_ptr(), _st()))
        sdyx = (s2 - mean * s1) * rstd  # sum dy xhat
        # dx = w rstd (dy - sum(dy) / m - xhat sum(dy xhat) / m),  xhat = (x - mean) rstd
        A = (wt * rstd).float().contiguous()
        B = (-wt * rstd * rstd * sdyx / m).float().contiguous()
        C = (wt * rstd * (-s1 / m + mean * rstd * sdyx / m)).float().contiguous()
        dx = torch.empty_like(x)
        _lib.check(L.cae_t_bn_affine(dy.data_ptr(), x.data_ptr(), n, c, h * w, A.data_ptr(), B.data_ptr(), C.data_ptr(),
                                     dx.data_ptr(), _st()))
        has_w, has_b = ctx.has
        return dx, (sdyx.float() if has_w else None), (s1.float() if has_b else None), None


def _batch_norm(bn: nn.BatchNorm2d, x: torch.Tensor) -> torch.Tensor:
    """bn(x) in training mode, with the running statistics updated as nn.BatchNorm2d does (unbiased variance, momentum or
    the cumulative average when momentum is None)."""
    y, mean, var = _BatchNormFn.apply(x, bn.weight, bn.bias, bn.eps)
    if bn.track_running_stats and bn.running_mean is not None:
        with torch.no_grad():
            bn.num_batches_tracked += 1
            mom = bn.momentum if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)
            cnt = x.numel() / x.shape[1]
            bn.running_mean.mul_(1.0 - mom).add_(mean.to(bn.running_mean.dtype), alpha=mom)
            bn.running_var.mul_(1.0 - mom).add_((var * (cnt / max(cnt - 1.0, 1.0))).to(bn.running_var.dtype), alpha=mom)
    return y


def _dense_weight(m) -> torch.Tensor:
    """The dense (groups = 1) weight of a grouped layer as a differentiable function of its parameter: block-diagonal
    embedding by one index_put (modules._ConvParams.dense_weight is its no-grad twin for the inference upload); the
    gradient of the grouped weight is the block diagonal of the dense kernels' weight gradient."""
    w = m.weight
    if m.groups == 1:
        return w
    g, k = m.groups, m.kernel_size
    cin_g, cout_g = m.in_channels // g, m.out_channels // g
    dev = w.device
    if m.transposed:  # (cin, cout_g, k, k) -> (cin, cout, k, k)
        rows = torch.arange(m.in_channels, device=dev)
        cols = (rows // cin_g)[:, None] * cout_g + torch.arange(cout_g, device=dev)[None, :]
        dense = w.new_zeros(m.in_channels, m.out_channels, k, k)
    else:             # (cout, cin_g, k, k) -> (cout, cin, k, k)
        rows = torch.arange(m.out_channels, device=dev)
        cols = (rows // cout_g)[:, None] * cin_g + torch.arange(cin_g, device=dev)[None, :]
        dense = w.new_zeros(m.out_channels, m.in_channels, k, k)
    return dense.index_put((rows[:, None].expand_as(cols), cols), w)


def _run_sequence(u, seq, x: torch.Tensor, synthesis: bool) -> torch.Tensor:
    """The modules of a unit's nn.Sequential, one operation each (the module order IS the reference's forward)."""
    from .modules import GDN, _ConvParams
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, _ConvParams):
            wd = _dense_weight(m)  # (grouped layers: the dense kernels on the block-diagonal embedding)
            if m is u.main:  # the strided layer
                if synthesis:
                    spec = LayerSpec(m.in_channels, m.out_channels, m.kernel_size, m.bias is not None, False)
                    x = SynthesisFn.apply(x, (spec,), *([wd] + ([m.bias] if m.bias is not None else [])))
                else:
                    x = _ConvS2Fn.apply(x, m.kernel_size, wd, m.bias)
            else:  # stride 1; a LeakyReLU / ReLU right behind it rides in the kernel's epilogue
                nxt = mods[i + 1] if i + 1 < len(mods) else None
                act = 1 if isinstance(nxt, nn.LeakyReLU) else (2 if isinstance(nxt, nn.ReLU) else 0)
                x = _ConvS1Fn.apply(x, synthesis, m.kernel_size, act, wd, m.bias)
                i += 1 if act else 0
        elif isinstance(m, nn.BatchNorm2d):
            x = _batch_norm(m, x) if m.training else m(x)
        elif isinstance(m, GDN):
            x = _GdnFn.apply(x, m.inverse, *_gdn_params(m, m.in_channels))
        elif isinstance(m, nn.LeakyReLU):
            x = torch.nn.functional.leaky_relu(x, m.negative_slope)
        elif isinstance(m, nn.ReLU):
            x = torch.relu(x)
        elif isinstance(m, nn.Dropout2d):
            x = m(x)  # (channel mask from torch's generator, as in the reference; the identity in eval mode)
        elif not isinstance(m, nn.Identity):
            raise NotImplementedError(f'{type(m).__name__} is not part of the compression path')
        i += 1
    return x


def _composed_track(units, x: torch.Tensor, synthesis: bool) -> torch.Tensor:
    """Tracks with residual units (y = model(res_model(x) + x), _autoencoders.py:168-174, :298-304) or batch norm:
    composed per operation."""
    from .modules import _ResidualUnit
    for u in units:
        if isinstance(u, _ResidualUnit):
            x = _run_sequence(u, u.model, _run_sequence(u, u.res_model, x, synthesis) + x, synthesis)
        else:
            x = _run_sequence(u, u.model, x, synthesis)
    return x


def _track_inputs(track, units, synthesis: bool):
    """-> (specs, flat tensor list) of a track; effective, padded GDN parameters stay in the autograd graph."""
    from .modules import _ResidualUnit
    specs, tensors = [], []
    if getattr(track, 'multiscale_analysis', False):
        raise NotImplementedError('training with multiscale colour layers is not built')
    for u in units:
        if (isinstance(u, _ResidualUnit) or u.main_bn_index is not None or u.pre_bn_index is not None or u.main.groups != 1
                or any(isinstance(m, nn.Dropout2d) and m.p > 0 for m in u.model)):
            return None, None  # composed per operation: _composed_track
        conv = u.main
        has_gdn = u.gdn is not None
        specs.append(LayerSpec(conv.in_channels, conv.out_channels, conv.kernel_size, conv.bias is not None, has_gdn,
                               act=u.act_code, has_pre=u.pre is not None))
        if u.pre is not None:  # LeakyReLU / ReLU units: the stride-1 convolution in front (same bias setting as the layer)
            tensors.append(u.pre.weight)
            if conv.bias is not None:
                tensors.append(u.pre.bias)
        tensors.append(conv.weight)
        if conv.bias is not None:
            tensors.append(conv.bias)
        if has_gdn:
            g = u.gdn
            c, cp = conv.out_channels, _pad32(conv.out_channels)
            beta = g.beta_reparam(g.beta)      # LowerBound gradient rule inside (entropy._LowerBoundFn)
            gamma = g.gamma_reparam(g.gamma)
            beta_p = torch.cat([beta, beta.new_ones(cp - c)]) if cp > c else beta
            gamma_p = torch.nn.functional.pad(gamma, (0, cp - c, 0, cp - c)) if cp > c else gamma
            tensors.extend([beta_p, gamma_p])
    return tuple(specs), tensors


def needs_grad(module: nn.Module, x: torch.Tensor) -> bool:
    """The differentiable track functions run when the module is in TRAINING mode and autograd is recording
    (train_cae_ms.py:183-187 puts the trainable modules in train(), the others in eval() under fixed_module's no_grad).
    In eval mode the inference kernels run whether or not autograd is enabled: the reference's codec.encode forgets
    torch.no_grad() (_autoencoders.py:539-555) and must keep its inference numerics."""
    if not (module.training and torch.is_grad_enabled()):
        return False
    return x.requires_grad or any(p.requires_grad for p in module.parameters())


def analysis_forward(track, x: torch.Tensor) -> torch.Tensor:
    dev = _lib.require_gpu()
    specs, tensors = _track_inputs(track, track._units(), False)
    x = x.to(device=dev, dtype=torch.float32)
    if specs is None:
        return _composed_track(track._units(), x, False)
    return AnalysisFn.apply(x, specs, *tensors)


def synthesis_forward(track, yq: torch.Tensor):
    dev = _lib.require_gpu()
    specs, tensors = _track_inputs(track, track._units(), True)
    yq = yq.to(device=dev, dtype=torch.float32)
    out = _composed_track(track._units(), yq, True) if specs is None else SynthesisFn.apply(yq, specs, *tensors)
    L = len(track._units())
    # (x_r list, fx_brg) as the reference's Synthesizer; intermediate features are not materialised while training
    return [out] + [None] * (L - 1), [None] * (L - 1) + [out]


# ---- optimisers and the training step (train_cae_ms.py) -----------------------------------------------------------

def setup_optim(model: Dict[str, nn.Module], trainable_modules: Sequence[str] = ('encoder', 'decoder', 'fact_ent'),
                learning_rate: float = 1e-4, aux_learning_rate: float = 1e-3, weight_decay: float = 0.0,
                algo=torch.optim.Adam, capturable: bool = False) -> Dict[str, torch.optim.Optimizer]:
    """One optimiser per trainable module; parameters whose name contains 'quantiles' or 'aux' go to a separate
    ``<module>_aux`` optimiser (train_cae_ms.py:584-641).  capturable: step counters on the device (no host read per
    step)."""
    opts: Dict[str, torch.optim.Optimizer] = {}
    extra = dict(capturable=True) if capturable else {}
    for k in trainable_modules:
        pars, aux = [], []
        for name, par in model[k].named_parameters():
            (aux if ('quantiles' in name.lower() or 'aux' in name.lower()) else pars).append(par)
        opts[k] = algo([dict(params=pars, lr=learning_rate, weight_decay=weight_decay)], **extra)
        if aux:
            opts[k + '_aux'] = algo([dict(params=aux, lr=aux_learning_rate, weight_decay=weight_decay)], **extra)
    return opts


_OPTIM_WS: Dict[torch.device, torch.Tensor] = {}


def fused_clip_adam(optimizers: Dict[str, torch.optim.Optimizer], max_norm: float = 1.0) -> bool:
    """`for opt: clip_grad_norm_(params, max_norm); opt.step(); opt.zero_grad()` (train_cae_ms.py:221-230) for ALL
    optimisers in two launches (cae_t_clip_adam), on the optimisers' own state tensors (exp_avg, exp_avg_sq, step), so
    `state_dict()` / checkpoints stay torch.optim.Adam's.  -> False (nothing done) unless every optimiser is a plain
    torch.optim.Adam (one parameter group, no amsgrad / maximize / capturable) over contiguous fp32 CUDA parameters;
    CAE_FUSED_OPTIM=0 disables it."""
    if os.environ.get('CAE_FUSED_OPTIM', '1') == '0' or len(optimizers) > 8:
        return False
    items = []  # (p, grad, exp_avg, exp_avg_sq, group index)
    groups = []
    for gi, opt in enumerate(optimizers.values()):
        if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
            return False
        g = opt.param_groups[0]
        if g.get('amsgrad') or g.get('maximize') or g.get('capturable') or g.get('differentiable') or \
                not isinstance(g['lr'], (int, float)):
            return False
        live = [p for p in g['params'] if p.grad is not None]
        for p in live:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous() or p.grad.dtype != torch.float32:
                return False
        groups.append((opt, g, live))
    for gi, (opt, g, live) in enumerate(groups):
        for p in live:
            st = opt.state[p]
            if len(st) == 0:  # as torch.optim.Adam initialises it lazily
                st['step'] = torch.tensor(0.0)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if not (st['exp_avg'].is_contiguous() and st['exp_avg_sq'].is_contiguous()) or st['step'].is_cuda:
                return False
            items.append((p, p.grad if p.grad.is_contiguous() else p.grad.contiguous(), st['exp_avg'], st['exp_avg_sq'], gi))
    if not items or len(items) > 64:
        return False
    n, G = len(items), len(groups)
    steps = []
    for opt, g, live in groups:
        steps.append(int(opt.state[live[0]]['step'].item()) + 1 if live else 1)
        for p in live:
            opt.state[p]['step'] += 1
    vp = lambda k: (ctypes.c_void_p * n)(*[it[k].data_ptr() for it in items])  # noqa: E731
    numel = (ctypes.c_int * n)(*[it[0].numel() for it in items])
    grp = (ctypes.c_int * n)(*[it[4] for it in items])
    fl = lambda vals: (ctypes.c_float * G)(*vals)  # noqa: E731
    chunks = sum((it[0].numel() + 2047) // 2048 for it in items)
    dev = items[0][0].device
    ws = _OPTIM_WS.get(dev)
    if ws is None or ws.numel() < chunks:
        ws = _OPTIM_WS[dev] = torch.empty(max(chunks, 1024), dtype=torch.float32, device=dev)
    _lib.check(_L().cae_t_clip_adam(
        n, vp(0), vp(1), vp(2), vp(3), numel, grp, G, fl([g['lr'] for _, g, _ in groups]),
        fl([g['betas'][0] for _, g, _ in groups]), fl([g['betas'][1] for _, g, _ in groups]), fl([g['eps'] for _, g, _ in groups]),
        fl([g['weight_decay'] for _, g, _ in groups]), fl([max_norm] * G), (ctypes.c_int * G)(*steps), ws.data_ptr(), ws.numel(),
        _st()))
    for opt, g, _ in groups:  # zero_grad(set_to_none=True)
        for p in g['params']:
            p.grad = None
    return True


def train_step(x: torch.Tensor, model, criterion, optimizers, forward_func=None, reducer: 'GradReducer' = None):
    """One iteration of the reference's hot loop (train_cae_ms.py:209-230).  -> loss_dict (detached scalars)"""
    from .criteria import setup_forward_func
    forward_func = forward_func or setup_forward_func()
    output = forward_func(x, model)
    loss_dict = criterion(inputs=x, outputs=output, net=model)
    loss = torch.mean(loss_dict['loss'])
    loss.backward()
    if 'entropy_loss' in loss_dict:
        torch.mean(loss_dict['entropy_loss']).backward()
    if reducer is not None:
        reducer.reduce()
    if not fused_clip_adam(optimizers, max_norm=1.0):
        for opt in optimizers.values():
            nn.utils.clip_grad_norm_(opt.param_groups[0]['params'], max_norm=1.0)
            opt.step()
            opt.zero_grad()
    return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in loss_dict.items()}


class GradReducer:
    """Data-parallel gradient averaging, one process per GPU (replaces nn.DataParallel's reduce-add to device 0,
    _autoencoders.py:517): the gradients are packed into a few flat fp32 buckets and all-reduced (RCCL over xGMI on GPUs,
    gloo in the CPU tests), averaged, and handed back as views of the bucket.  OVERLAPPED with the backward pass: a
    post-accumulate hook per parameter counts a bucket down and launches its all-reduce (async) the moment its last
    gradient exists -- the backward runs decoder, entropy model, encoder, so the decoder's bucket is on the wire while the
    encoder's gradients are still being computed; ``reduce()`` (between ``backward()`` and the optimiser step) launches
    what is left (parameters without a gradient count as zero), waits and averages.  The buckets are persistent: no
    per-step concatenation.  The model holds about 2 M parameters (8 MB): latency-bound, a single ring."""

    def __init__(self, params: Sequence[torch.Tensor], bucket_bytes: int = 2 << 20, overlap: bool = True):
        self.params = [p for p in params if p.requires_grad]
        self.buckets: List[List[torch.Tensor]] = [[]]
        size = 0
        for p in self.params:
            nbytes = p.numel() * 4
            if size + nbytes > bucket_bytes and self.buckets[-1]:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append(p)
            size += nbytes
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._views: List[Optional[List[torch.Tensor]]] = [None] * len(self.buckets)
        self._works: List = [None] * len(self.buckets)
        self._pending = [len(b) for b in self.buckets]
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self.launched_in_backward = 0  # (statistics: buckets whose all-reduce started from a hook)
        self._hooks = []
        if overlap and hasattr(torch.Tensor, 'register_post_accumulate_grad_hook'):
            for p in self.params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    @staticmethod
    def _active() -> bool:
        import torch.distributed as dist
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def _on_grad(self, p):
        if not self._active():
            return
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0 and self._works[i] is None:
            self._launch(i)
            self.launched_in_backward += 1

    @torch.no_grad()
    def _launch(self, i: int):
        import torch.distributed as dist
        bucket = self.buckets[i]
        if self._flat[i] is None or self._flat[i].device != bucket[0].device:
            self._flat[i] = torch.zeros(sum(p.numel() for p in bucket), dtype=torch.float32, device=bucket[0].device)
            views, off = [], 0
            for p in bucket:
                views.append(self._flat[i][off:off + p.numel()].view_as(p))
                off += p.numel()
            self._views[i] = views
        have = [(v, p.grad) for v, p in zip(self._views[i], bucket) if p.grad is not None and p.grad.data_ptr() != v.data_ptr()]
        for v, p in zip(self._views[i], bucket):
            if p.grad is None:
                v.zero_()
        if have:
            torch._foreach_copy_([v for v, _ in have], [g.to(torch.float32) for _, g in have])
        self._works[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, async_op=True)

    @torch.no_grad()
    def reduce(self):
        import torch.distributed as dist
        if not self._active():
            return
        world = dist.get_world_size()
        for i in range(len(self.buckets)):
            if self._works[i] is None:
                self._launch(i)
        for i, bucket in enumerate(self.buckets):
            self._works[i].wait()
            self._flat[i].div_(world)
            for v, p in zip(self._views[i], bucket):
                p.grad = v if p.dtype == torch.float32 else v.to(p.dtype)
            self._works[i] = None
            self._pending[i] = len(bucket)
