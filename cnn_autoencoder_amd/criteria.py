"""Evaluation side of the reference's training objective (config 5's `valid` loop): the forward function of
``models/tasks/_taskutils.py:95-108`` and the rate / distortion terms of ``models/criteria`` -- ``RateLoss`` and
``DistMSELoss`` (``_ratedist.py:45-63``) assembled as ``GeneralLoss`` does (``_lossutils.py:54-72,100-109``):
``loss = lambda * 255^2 * MSE(x_r, x) + (-sum log2 p_y / (B H W))`` plus the reported ``entropy_loss``.

Under ``torch.no_grad()`` the analysis / synthesis tracks and the eval-mode density run on the inference kernels; with
autograd recording (``train.train_step``) the tracks switch to the training kernels with hand-written backward
(``train.py``) and the entropy model adds its uniform noise (train mode).  Classifier / segmentation heads, penalty
terms and the MS-SSIM / pyramid distortions of the reference are outside the hot path and not built.
"""
from __future__ import annotations

from typing import Dict, Sequence, Union

import torch
import torch.nn as nn


def setup_forward_func(enabled_modules: Sequence[str] = ('encoder', 'fact_ent', 'decoder')):
    """decorate_trainable_modules / forward_func of the reference for the modules of the hot path; disabled modules
    are identities exactly as there (`_taskutils.py:40-80`)."""
    enabled = set(enabled_modules)
    unknown = enabled - {'encoder', 'fact_ent', 'decoder'}
    if unknown:
        raise NotImplementedError(f'modules outside the compression path are not built: {sorted(unknown)}')

    def forward_func(x, model) -> Dict:
        y = model['encoder'](x) if 'encoder' in enabled else x
        y_q, p_y = model['fact_ent'](y) if 'fact_ent' in enabled else (y, None)
        x_r, fx_brg = model['decoder'](y_q) if 'decoder' in enabled else (y_q, None)
        return dict(x_r=x_r, fx_brg=fx_brg, y=y, y_q=y_q, p_y=p_y, t_pred=None, t_aux_pred=None, s_pred=None,
                    s_aux_pred=None)

    return forward_func


class RateLoss:
    """_ratedist.py:45-54."""

    def __init__(self, **kwargs):
        pass

    def __call__(self, x, p_y, **kwargs):
        rate_loss = -torch.sum(torch.log2(p_y)) / (x.size(0) * x.size(2) * x.size(3))
        return dict(rate_loss=rate_loss)


class DistMSELoss:
    """_ratedist.py:57-63."""

    def __init__(self, **kwargs):
        self._dist_loss = nn.MSELoss()

    def __call__(self, x, x_r, **kwargs):
        return dict(dist=[self._dist_loss(x_r[0], x.to(x_r[0].device))])


class GeneralLoss(nn.Module):
    """_lossutils.py:5-109 restricted to dist_loss_type='MSE' | None and rate_loss_type='Rate' | None."""

    def __init__(self, dist_loss_type='MSE', rate_loss_type='Rate', penalty_loss_type=None, class_loss_type=None,
                 distortion_lambda: Union[float, Sequence[float]] = 0.1, **kwargs):
        super().__init__()
        if dist_loss_type not in (None, 'MSE') or rate_loss_type not in (None, 'Rate'):
            raise NotImplementedError('only the MSE distortion and the Rate term are built')
        for name, v in (('penalty_loss_type', penalty_loss_type), ('class_loss_type', class_loss_type)):
            if v is not None and str(v).lower() != 'none':
                raise NotImplementedError(f'{name}={v!r} is outside the compression path')
        self.dist_loss = DistMSELoss() if dist_loss_type else None
        self.rate_loss = RateLoss() if rate_loss_type else None
        self._multiplier = 255 ** 2
        self._distortion_lambda = list(distortion_lambda) if isinstance(distortion_lambda, (list, tuple)) else [distortion_lambda]

    def forward(self, inputs, outputs, targets=None, net=None, **kwargs):
        loss_dict = {'loss': 0, 'channel_e': torch.LongTensor([-1])}
        if self.dist_loss is not None:
            loss_dict.update(self.dist_loss(x=inputs, x_r=outputs['x_r']))
            loss_dict['dist'] = [self._multiplier * d for d in loss_dict['dist']]
            loss_dict['dist_loss'] = sum(d * w for d, w in zip(loss_dict['dist'], self._distortion_lambda))
            loss_dict['loss'] = loss_dict['loss'] + loss_dict['dist_loss']
        if self.rate_loss is not None:
            loss_dict.update(self.rate_loss(x=inputs, p_y=outputs['p_y']))
            fe = net['fact_ent']
            loss_dict['entropy_loss'] = getattr(fe, 'module', fe).loss()
            loss_dict['loss'] = loss_dict['loss'] + loss_dict['rate_loss']
        return loss_dict


def setup_loss(criterion: str, **kwargs) -> GeneralLoss:
    """``models/criteria/_lossutils.py:112-151``: the criterion NAME selects the terms ('RateMSE' = the reference's
    default, ``utils/args/_critargs.py:42``).  Built: the Rate term and the MSE distortion; the names of the terms outside
    the compression path (MS-SSIM / multiscale distortions, penalties, classification losses) raise."""
    name = criterion.lower()
    rate = 'Rate' if 'rate' in name else None
    if 'mse' in name:
        dist = 'MSE'
    elif 'msssim' in name or 'ms-ssim' in name:
        raise NotImplementedError('the MS-SSIM distortion is outside the compression hot path')
    else:
        dist = None
    # (the reference tests `'pa' in name` / `'ce' in name` as plain substrings; those terms are not built here)
    for key in ('multiscale', 'penalty', 'crossentropy', 'bce', 'weighted'):
        if key in name:
            raise NotImplementedError(f'criterion {criterion!r}: the {key} term is outside the compression hot path')
    return GeneralLoss(dist, rate, 'none', None, **kwargs)
