"""CPU restatement of the TRAINING step (test infrastructure only: tests/, smoke(), bench cpu_baseline).

torch-CPU autograd of the restatement in cae_oracle.py, i.e. of what the reference differentiates
(train_cae_ms.py:209-219: forward_func of models/tasks/_taskutils.py:95-108, GeneralLoss of
models/criteria/_lossutils.py:54-109 with RateLoss / DistMSELoss of _ratedist.py:45-63), plus the two gradient rules
that live in compressai and therefore are "parity unpinned" (SURVEY Appendix A.1 / A.2):
  * LowerBound: forward max(x, bound); backward passes the gradient where x >= bound or gradient < 0;
  * NonNegativeParametrizer: lower_bound(p, sqrt(minimum + pedestal))^2 - pedestal.
The conv / conv-transpose parts are reference-pinned through cae_oracle (golden fixtures of the reference's own modules).

`bf16=True` reproduces the ROUNDING POINTS of the HIP training path (BASELINE config 5: bf16 convolutions, fp32 GDN):
activations and weights enter a convolution rounded to bf16, the gradient with respect to a convolution's output is
rounded to bf16 before the data / weight gradients are formed; accumulation, GDN, losses stay fp32.  With it the only
difference to the kernels is summation order; without it this is the plain fp32 reference.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import cae_oracle as O


class LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)).to(g.dtype) * g, None


def lower_bound(x: torch.Tensor, bound: float) -> torch.Tensor:
    return LowerBoundFn.apply(x, torch.tensor([bound], dtype=x.dtype))


def nonneg_reparam(p: torch.Tensor, minimum: float) -> torch.Tensor:
    return lower_bound(p, (minimum + O.PEDESTAL) ** 0.5) ** 2 - O.PEDESTAL


def gdn(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor, inverse: bool, beta_min: float = 1e-6) -> torch.Tensor:
    C = x.shape[1]
    b = nonneg_reparam(beta, beta_min)
    g = nonneg_reparam(gamma, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x ** 2, g, b)
    return x * (torch.sqrt(norm) if inverse else torch.rsqrt(norm))


class _Round(torch.autograd.Function):
    """value rounded to bf16, gradient passed through"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _GradRound(torch.autograd.Function):
    """identity whose incoming gradient is rounded to bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _r(x, bf16):
    return _Round.apply(x) if bf16 else x


def _g(x, bf16):
    return _GradRound.apply(x) if bf16 else x


def _act(x: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    if act == 'LeakyReLU':
        return F.leaky_relu(x, 0.01)
    if act == 'ReLU':
        return F.relu(x)
    return x


def analysis(x: torch.Tensor, layers: Sequence[dict], bf16: bool = True, act: Optional[str] = None) -> torch.Tensor:
    """layers[i] = {'weight', 'bias'?, 'beta'?, 'gamma'?, 'pre_weight'?, 'pre_bias'?} (stored parameters, leaf tensors
    with requires_grad).  LeakyReLU / ReLU units (`act`; DownsamplingUnit, _autoencoders.py:62-92): a stride-1 reflect
    convolution + activation in front of the strided layer, the activation after it; the last unit has neither."""
    fx = x
    for i, L in enumerate(layers):
        if L.get('pre_weight') is not None:
            k = L['pre_weight'].shape[-1]
            u = F.conv2d(F.pad(_r(fx, bf16), (k // 2,) * 4, mode='reflect'), _r(L['pre_weight'], bf16), L.get('pre_bias'))
            fx = _act(_g(u, bf16), act)
        fx = _g(O.reflect_conv_s2(_r(fx, bf16), _r(L['weight'], bf16), L.get('bias')), bf16)
        if L.get('beta') is not None:
            fx = gdn(fx, L['beta'], L['gamma'], False)
        elif act is not None and i < len(layers) - 1:
            fx = _act(fx, act)
    return fx


def synthesis(y: torch.Tensor, layers: Sequence[dict], bf16: bool = True, act: Optional[str] = None) -> torch.Tensor:
    fx = y
    for i, L in enumerate(layers):
        if L.get('pre_weight') is not None:  # ConvTranspose2d(cin, cin, k, stride 1, padding k//2) (UpsamplingUnit :187-202)
            k = L['pre_weight'].shape[-1]
            u = F.conv_transpose2d(_r(fx, bf16), _r(L['pre_weight'], bf16), L.get('pre_bias'), stride=1, padding=k // 2)
            fx = _act(_g(u, bf16), act)
        fx = _g(O.deconv_s2(_r(fx, bf16), _r(L['weight'], bf16), L.get('bias')), bf16)
        if L.get('beta') is not None:
            fx = gdn(fx, L['beta'], L['gamma'], True)
        elif act is not None and i < len(layers) - 1:
            fx = _act(fx, act)
    return fx


def _bn(x: torch.Tensor, bn: Optional[dict]) -> torch.Tensor:
    """nn.BatchNorm2d in training mode (batch statistics): bn = {'weight', 'bias', 'eps'} or None"""
    if bn is None:
        return x
    return F.batch_norm(x, None, None, bn.get('weight'), bn.get('bias'), True, 0.0, bn.get('eps', 1e-5))


def residual_track(x: torch.Tensor, units: Sequence[dict], synthesis: bool, bf16: bool = True) -> torch.Tensor:
    """Residual units (ResidualDownsamplingUnit / ResidualUpsamplingUnit, _autoencoders.py:104-174, :230-304):
    y = model(res_model(x) + x).  units[i] = {'stages': [{'weight', 'bias'?, 'beta'?, 'gamma'?, 'act'?}], 'post_act'?,
    'weight', 'bias'?, 'beta'?, 'gamma'?, 'act'?}: res_model = stride-1 convolutions cin -> cin (analysis: reflect padding;
    synthesis: ConvTranspose2d(stride 1, padding k//2)) each followed by GDN / IGDN or an activation, model = [activation]
    + the strided layer + GDN / IGDN or activation.  Rounding points of the HIP path (`bf16`): operands of every convolution,
    the gradient at every convolution's output, the output of every GDN, and the gradient an analysis-side convolution hands to
    its input (folded from the extended domain into bf16).  Optional 'bn' entries ({'weight', 'bias', 'eps'}) behind a
    convolution: BatchNorm2d with batch statistics (:72-73, :87-88); units with 'residual': False are the plain
    DownsamplingUnit / UpsamplingUnit (their stride-1 pre-convolution as the one stage, no residual sum)."""
    fx = x
    for U in units:
        r = fx
        for S in U['stages']:
            k = S['weight'].shape[-1]
            rin = _r(r, bf16) if synthesis else _r(_g(r, bf16), bf16)
            if synthesis:
                u = F.conv_transpose2d(rin, _r(S['weight'], bf16), S.get('bias'), stride=1, padding=k // 2)
            else:
                u = F.conv2d(F.pad(rin, (k // 2,) * 4, mode='reflect'), _r(S['weight'], bf16), S.get('bias'))
            u = _bn(_g(u, bf16), S.get('bn'))
            if S.get('beta') is not None:
                r = _r(gdn(_g(u, bf16), S['beta'], S['gamma'], synthesis), bf16)  # (the GDN backward kernel emits bf16)
            else:
                r = _act(u, S.get('act'))
        if U.get('residual', True):
            r = _act(r + fx, U.get('post_act'))
        if synthesis:
            y = _g(O.deconv_s2(_r(r, bf16), _r(U['weight'], bf16), U.get('bias')), bf16)
        else:
            y = _g(O.reflect_conv_s2(_r(_g(r, bf16), bf16), _r(U['weight'], bf16), U.get('bias')), bf16)
        y = _bn(y, U.get('bn'))
        if U.get('beta') is not None:
            y = _r(gdn(_g(y, bf16), U['beta'], U['gamma'], synthesis), bf16)
        else:
            y = _act(y, U.get('act'))
        fx = y
    return fx


def entropy_forward(params: dict, y: torch.Tensor, noise: Optional[torch.Tensor], n_filters: int, form: str = 'plain',
                    bound: float = 1e-9):
    """EntropyBottleneck.forward in train mode: y + noise, likelihood with the LowerBound rule.  noise: like y."""
    perm = list(range(y.dim()))
    perm[0], perm[1] = 1, 0
    v = y.permute(*perm).contiguous()
    shape = v.shape
    v = v.reshape(shape[0], 1, -1)
    if noise is not None:
        v = v + noise.permute(*perm).reshape(shape[0], 1, -1)

    def logits(t):
        for i in range(n_filters + 1):
            t = torch.matmul(F.softplus(params[f'_matrix{i}']), t) + params[f'_bias{i}']
            if i < n_filters:
                t = t + torch.tanh(params[f'_factor{i}']) * torch.tanh(t)
        return t
    lower, upper = logits(v - 0.5), logits(v + 0.5)
    if form == 'plain':
        lik = torch.sigmoid(upper) - torch.sigmoid(lower)
    else:
        s = -torch.sign(lower + upper).detach()
        lik = torch.abs(torch.sigmoid(s * upper) - torch.sigmoid(s * lower))
    lik = lower_bound(lik, bound)
    return v.reshape(shape).permute(*perm).contiguous(), lik.reshape(shape).permute(*perm).contiguous()


def aux_loss(params: dict, n_filters: int, target: torch.Tensor) -> torch.Tensor:
    """EntropyBottleneck.loss(): |logits(quantiles) - target| summed, gradient to `quantiles` only"""
    t = params['quantiles']
    for i in range(n_filters + 1):
        t = torch.matmul(F.softplus(params[f'_matrix{i}'].detach()), t) + params[f'_bias{i}'].detach()
        if i < n_filters:
            t = t + torch.tanh(params[f'_factor{i}'].detach()) * torch.tanh(t)
    return torch.abs(t - target).sum()


def rd_loss(x: torch.Tensor, x_r: torch.Tensor, p_y: torch.Tensor, distortion_lambda: float):
    """GeneralLoss: rate = -sum log2 p / (B H W), dist = 255^2 MSE; loss = rate + lambda dist"""
    rate = -torch.sum(torch.log2(p_y)) / (x.size(0) * x.size(2) * x.size(3))
    dist = 255 ** 2 * F.mse_loss(x_r, x)
    return rate + distortion_lambda * dist, rate, dist
