"""Seeded synthetic workloads: tiles and model parameters.

The reference ships no checkpoint, config or dataset, so every
parity fixture and bench input is generated from the seeds fixed in SURVEY.md §8(d).
Everything here is numpy-RNG driven (``default_rng`` streams are stable across numpy
versions) so that the same parameters can be rebuilt on the GPU box without shipping
weights.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

REPARAM_OFFSET = 2.0 ** -18
PEDESTAL = REPARAM_OFFSET ** 2

LOW_RATE = dict(latent_gain=0.5, prior_scale=0.2)  # synthetic_state(cfg, **LOW_RATE): ~0.9 bpp on histology tiles (96 % zero symbols)

CANONICAL = dict(channels_org=3, channels_net=128, channels_bn=192, compression_level=4,
                 channels_expansion=1, kernel_size=3, groups=False, batch_norm=False,
                 dropout=0.0, bias=False, use_residual=False, act_layer_type='GDN',
                 K=4, r=3)


def uniform_tiles(n: int, h: int, w: Optional[int] = None, c: int = 3, seed: int = 1234) -> np.ndarray:
    """SURVEY §8(d) `uniform`: worst-case-entropy tiles, (n,h,w,c) uint8."""
    w = h if w is None else w
    return np.random.default_rng(seed).integers(0, 256, (n, h, w, c), dtype=np.uint8)


def histo_tile(h: int, tile_index: int = 0, w: Optional[int] = None) -> np.ndarray:
    """SURVEY §8(d) `histo`: procedural H&E-like tile, (h,w,3) uint8, rng 2024+tile_index."""
    from scipy.ndimage import gaussian_filter
    w = h if w is None else w
    rng = np.random.default_rng(2024 + tile_index)
    img = np.empty((h, w, 3), dtype=np.float32)
    img[:] = (240, 240, 240)
    tissue = gaussian_filter(rng.standard_normal((h, w)).astype(np.float32), sigma=max(h, w) / 16) > 0
    blend = gaussian_filter(rng.standard_normal((h, w)).astype(np.float32), sigma=max(max(h, w) / 64, 1.0))
    blend = (blend - blend.min()) / max(float(blend.max() - blend.min()), 1e-6)
    eosin = np.array((233, 163, 199), dtype=np.float32)
    stroma = np.array((245, 205, 225), dtype=np.float32)
    col = eosin[None, None, :] * blend[..., None] + stroma[None, None, :] * (1 - blend[..., None])
    img[tissue] = col[tissue]
    nuclei = (gaussian_filter(rng.standard_normal((h, w)).astype(np.float32), sigma=3) * 3 * math.sqrt(4 * math.pi)
              > 1.2) & tissue
    img[nuclei] = (70, 40, 130)
    img += rng.normal(0, 3, img.shape).astype(np.float32)
    return np.clip(img, 0, 255).astype(np.uint8)


def histo_tiles(n: int, h: int, first_index: int = 0) -> np.ndarray:
    return np.stack([histo_tile(h, first_index + i) for i in range(n)])


def mnist_like(n: int, seed: int = 7) -> np.ndarray:
    """cfg1 stand-in: (n,32,32,1) uint8 blobs (28x28 digit-like strokes padded by 2)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((n, 32, 32, 1), dtype=np.uint8)
    yy, xx = np.mgrid[0:28, 0:28]
    for i in range(n):
        img = np.zeros((28, 28), dtype=np.float32)
        for _ in range(rng.integers(2, 5)):
            cy, cx = rng.uniform(6, 22, 2)
            sy, sx = rng.uniform(1.5, 5, 2)
            img += np.exp(-(((yy - cy) / sy) ** 2 + ((xx - cx) / sx) ** 2))
        img = np.clip(img / max(img.max(), 1e-6) * 255, 0, 255)
        out[i, 2:30, 2:30, 0] = img.astype(np.uint8)
    return out


def _xavier_bound(cin: int, cout: int, k: int) -> float:
    # reference initialize_weights: xavier_uniform_(gain=sqrt(2/1.01))  (_autoencoders.py:37-42)
    gain = math.sqrt(2 / 1.01)
    return gain * math.sqrt(6.0 / ((cin + cout) * k * k))


def _nonneg_init(x: np.ndarray) -> np.ndarray:
    return np.sqrt(np.maximum(x + PEDESTAL, PEDESTAL)).astype(np.float32)


def synthetic_state(cfg: Dict, seed: int = 0, stress: bool = False, latent_gain: float = 1.0,
                    prior_scale: float = 10.0) -> Dict:
    """Checkpoint-shaped dict: cfg keys + 'encoder'/'decoder'/'fact_ent' state dicts.

    Conv weights follow the reference init distribution; GDN effective beta in U(0.5,1.5) and
    gamma = 0.1 I + U(0,0.02) so the off-diagonal contraction is exercised; entropy-model
    parameters follow the EntropyBottleneck init (biases U(-.5,.5)).
    ``stress``: last analysis conv x40 so latents leave the CDF support (bypass path).
    ``latent_gain`` / ``prior_scale``: last analysis conv x latent_gain and the entropy model initialised with
    init_scale = prior_scale -- small values of both stand in for a TRAINED low-rate model (most symbols zero under a
    narrow prior, ~0.9 bpp instead of 4: LOW_RATE), the other end of what the host range coder sees; the defaults give the
    near-worst-case 4 bpp of random weights under the wide initial prior.
    """
    rng = np.random.default_rng(seed)
    L = cfg['compression_level']
    k = cfg.get('kernel_size', 3)
    c_org, c_net, c_bn = cfg['channels_org'], cfg['channels_net'], cfg['channels_bn']
    bias = cfg.get('bias', False)
    gdn = cfg.get('act_layer_type') == 'GDN'

    def u(shape, b):
        return torch.from_numpy(rng.uniform(-b, b, shape).astype(np.float32))

    act = cfg.get('act_layer_type')
    pre = act in ('LeakyReLU', 'ReLU')  # units carry a stride-1 conv + activation in front (model.0, model.1)
    enc, dec = {}, {}
    cin = c_org
    for i in range(L):
        cout = c_net if i < L - 1 else c_bn
        main = 0
        if pre and i < L - 1:
            enc[f'analysis_track.{i}.model.0.weight'] = u((cin, cin, k, k), _xavier_bound(cin, cin, k))
            if bias:
                enc[f'analysis_track.{i}.model.0.bias'] = u((cin,), 0.05)
            main = 2
        w = u((cout, cin, k, k), _xavier_bound(cin, cout, k))
        if stress and i == L - 1:
            w = w * 40.0
        if latent_gain != 1.0 and i == L - 1:
            w = w * float(latent_gain)
        enc[f'analysis_track.{i}.model.{main}.weight'] = w
        if bias:
            enc[f'analysis_track.{i}.model.{main}.bias'] = u((cout,), 0.05) if pre else torch.full((cout,), 0.01)
        if gdn and i < L - 1:
            beta = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
            gamma = (0.1 * np.eye(cout) + rng.uniform(0, 0.02, (cout, cout))).astype(np.float32)
            enc[f'analysis_track.{i}.model.1.beta'] = torch.from_numpy(_nonneg_init(beta))
            enc[f'analysis_track.{i}.model.1.gamma'] = torch.from_numpy(_nonneg_init(gamma))
        cin = cout
    cin = c_bn
    for i in range(L):
        cout = c_net if i < L - 1 else c_org
        main = 0
        if pre and i < L - 1:
            dec[f'synthesis_track.{i}.model.0.weight'] = u((cin, cin, k, k), _xavier_bound(cin, cin, k))
            if bias:
                dec[f'synthesis_track.{i}.model.0.bias'] = u((cin,), 0.05)
            main = 2
        dec[f'synthesis_track.{i}.model.{main}.weight'] = u((cin, cout, k, k), _xavier_bound(cin, cout, k))
        if bias:
            dec[f'synthesis_track.{i}.model.{main}.bias'] = u((cout,), 0.05) if pre else torch.full((cout,), 0.01)
        if gdn and i < L - 1:
            beta = rng.uniform(0.5, 1.5, (cout,)).astype(np.float32)
            gamma = (0.1 * np.eye(cout) + rng.uniform(0, 0.02, (cout, cout))).astype(np.float32)
            dec[f'synthesis_track.{i}.model.1.beta'] = torch.from_numpy(_nonneg_init(beta))
            dec[f'synthesis_track.{i}.model.1.gamma'] = torch.from_numpy(_nonneg_init(gamma))
        cin = cout

    K, r = cfg.get('K', 4), cfg.get('r', 3)
    filters = (1,) + (r,) * K + (1,)
    init_scale = float(prior_scale)
    scale = init_scale ** (1 / (K + 1))
    fe = {}
    for i in range(K + 1):
        init = float(np.log(np.expm1(1 / scale / filters[i + 1])))
        fe[f'_matrix{i}'] = torch.full((c_bn, filters[i + 1], filters[i]), init)
        fe[f'_bias{i}'] = u((c_bn, filters[i + 1], 1), 0.5)
        if i < K:
            fe[f'_factor{i}'] = torch.zeros(c_bn, filters[i + 1], 1)
    fe['quantiles'] = torch.tensor([-init_scale, 0.0, init_scale]).repeat(c_bn, 1, 1)
    state = dict(cfg)
    state.update(encoder=enc, decoder=dec, fact_ent=fe)
    return state
