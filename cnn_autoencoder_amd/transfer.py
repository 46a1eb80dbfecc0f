"""Checkpoint key mapping between compressai's ``bmshj2018-factorized`` layout and this codec's.

Same job as the reference's ``scripts/transfer_weights.py`` (key tables at :5-47, index arithmetic
:50-69): compressai numbers the modules of ``g_a`` / ``g_s`` flat (conv 0, GDN 1, conv 2, ...), the CAE
classes nest them as ``<track>.{N // 2}.model.{N % 2}``; old zoo files name the entropy parameters
``_matrices.{i}`` / ``_biases.{i}`` / ``_factors.{i}`` where the modules here (and recent compressai) use
``_matrix{i}`` / ``_bias{i}`` / ``_factor{i}``.  Tensors are passed through untouched.
"""
from __future__ import annotations

import re
from typing import Dict

import torch

_TRACKS = (('g_a.', 'encoder', 'analysis_track.'), ('g_s.', 'decoder', 'synthesis_track.'))
_OLD_EB = (('_matrices.', '_matrix'), ('_biases.', '_bias'), ('_factors.', '_factor'))


def compressai_to_cae(src: Dict[str, torch.Tensor]) -> Dict[str, Dict[str, torch.Tensor]]:
    """flat compressai state dict -> {'encoder': ..., 'decoder': ..., 'fact_ent': ...} state dicts."""
    out = {'encoder': {}, 'decoder': {}, 'fact_ent': {}}
    for key, w in src.items():
        for prefix, part, track in _TRACKS:
            if key.startswith(prefix):
                m = re.match(r'(\d+)\.(.+)$', key[len(prefix):])
                if m is None:
                    raise KeyError(f'unexpected key {key!r}')
                n = int(m.group(1))
                out[part][f'{track}{n // 2}.model.{n % 2}.{m.group(2)}'] = w
                break
        else:
            if key.startswith('entropy_bottleneck.'):
                k = key[len('entropy_bottleneck.'):]
                for old, new in _OLD_EB:
                    if k.startswith(old):
                        k = new + k[len(old):]
                out['fact_ent'][k] = w
            # anything else (hyper-prior branches of other zoo models, ...) has no counterpart
    return out


def cae_to_compressai(ckpt: Dict) -> Dict[str, torch.Tensor]:
    """checkpoint with 'encoder' / 'decoder' / 'fact_ent' state dicts -> flat compressai state dict."""
    out = {}
    for prefix, part, track in _TRACKS:
        for key, w in ckpt.get(part, {}).items():
            m = re.match(re.escape(track) + r'(\d+)\.model\.(\d+)\.(.+)$', key)
            if m is None:
                raise KeyError(f'unexpected key {key!r}')
            out[f'{prefix}{2 * int(m.group(1)) + int(m.group(2))}.{m.group(3)}'] = w
    for key, w in ckpt.get('fact_ent', {}).items():
        out['entropy_bottleneck.' + key] = w
    return out


def factorized_config(src: Dict[str, torch.Tensor]) -> Dict:
    """Model hyper-parameters implied by a bmshj2018-factorized state dict (shapes of g_a)."""
    w0 = src['g_a.0.weight']
    idx = sorted({int(k.split('.')[1]) for k in src if k.startswith('g_a.') and k.endswith('.weight')})
    last = src[f'g_a.{idx[-1]}.weight']
    eb_filters = [v.shape[1] for k, v in sorted(src.items()) if re.match(r'entropy_bottleneck\._(matrices\.|matrix)\d+$', k)]
    return dict(channels_org=int(w0.shape[1]), channels_net=int(w0.shape[0]), channels_bn=int(last.shape[0]),
                compression_level=len(idx), channels_expansion=1, kernel_size=int(w0.shape[-1]), groups=False,
                batch_norm=False, dropout=0.0, bias='g_a.0.bias' in src, use_residual=False, act_layer_type='GDN',
                K=len(eb_filters) - 1, r=int(eb_filters[0]) if eb_filters else 3)
