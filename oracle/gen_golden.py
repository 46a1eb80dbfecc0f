#!/usr/bin/env python3
"""Generate golden fixtures under tests/golden/ by running the REFERENCE's own classes.

Run only in the build container (``/root/reference`` does not exist on the GPU box):

    python oracle/gen_golden.py

What is executed from the reference: ``src/models/tasks/_autoencoders.py`` loaded by
file path, with the third-party names that are absent from this image
(``compressai``, ``numcodecs``) stubbed in ``sys.modules``.  The
reference's ``Analyzer`` / ``Synthesizer`` / ``DownsamplingUnit`` / ``UpsamplingUnit``
/ ``initialize_weights`` therefore run unmodified; the stub ``GDN`` is OUR restatement
(oracle/cae_oracle.py) because compressai's is absent, so in the GDN fixtures the
conv / conv-transpose layers and the unit wiring are reference-pinned while the GDN
arithmetic is "parity unpinned".  Fixtures with ``act=None`` are reference-pinned end
to end.

Only data (inputs, weights, expected outputs) is written; no reference source is copied.
"""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import cae_oracle as O  # noqa: E402
from cnn_autoencoder_amd import synth  # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
REF = '/root/reference/src/models/tasks/_autoencoders.py'


class StubGDN(nn.Module):
    """Stand-in for compressai.layers.GDN with the same ctor signature and parameter names."""

    def __init__(self, in_channels, inverse=False, beta_min=1e-6, gamma_init=0.1):
        super().__init__()
        self.inverse = bool(inverse)
        self.beta_min = float(beta_min)
        b, g = O.gdn_init_params(in_channels, gamma_init)
        self.beta = nn.Parameter(b)
        self.gamma = nn.Parameter(g)

    def forward(self, x):
        return O.gdn_forward(x, self.beta, self.gamma, self.inverse, self.beta_min)


def load_reference():
    for name in ['compressai', 'compressai.layers', 'compressai.entropy_models', 'numcodecs',
                 'numcodecs.abc', 'numcodecs.compat']:
        sys.modules[name] = types.ModuleType(name)
    sys.modules['compressai'].ans = None
    sys.modules['compressai.layers'].GDN = StubGDN
    sys.modules['compressai.entropy_models'].EntropyBottleneck = object
    sys.modules['numcodecs.abc'].Codec = object
    sys.modules['numcodecs.compat'].ndarray_copy = None
    sys.modules['numcodecs.compat'].ensure_contiguous_ndarray = None
    spec = importlib.util.spec_from_file_location('ref_autoencoders', REF)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def layer_outputs(seq, x):
    outs = []
    fx = x
    for unit in seq:
        fx = unit(fx)
        outs.append(fx)
    return outs


def stats(t):
    t = t.detach().double()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def make_case(ref, name, cfg, tile, seed, store_weights, store_layers):
    """Run the reference Analyzer/Synthesizer on `tile` with synthetic_state(cfg, seed)."""
    state = synth.synthetic_state(cfg, seed=seed)
    kw = {k: cfg[k] for k in ['channels_org', 'channels_net', 'channels_bn', 'compression_level',
                              'channels_expansion', 'kernel_size', 'groups', 'batch_norm', 'dropout',
                              'bias', 'use_residual', 'act_layer_type']}
    enc = ref.Analyzer(**kw)
    dec = ref.Synthesizer(**kw)
    missing = enc.load_state_dict(state['encoder'], strict=True)
    missing = dec.load_state_dict(state['decoder'], strict=True)
    enc.eval()
    dec.eval()
    out = {}
    with torch.no_grad():
        x = O.tile_to_input(tile)
        enc_outs = layer_outputs(enc.analysis_track, x)
        y = enc(x)
        assert torch.equal(y, enc_outs[-1])
        # entropy model is ours (unpinned): quantise with the oracle's medians (all zero at init)
        yq = torch.round(y)
        x_r, brg = dec(yq)
        dec_outs = layer_outputs(dec.synthesis_track, yq)
        assert torch.equal(x_r[0], dec_outs[-1]) and torch.equal(brg[-1], dec_outs[-1])
        assert all(t is None for t in x_r[1:])
        u8 = O.output_to_tile(x_r[0][0])
    out['tile'] = tile
    out['y'] = y.numpy()
    out['x_r'] = x_r[0].numpy()
    out['x_r_u8'] = u8
    for i, t in enumerate(enc_outs):
        out[f'enc_stats_{i}'] = stats(t)
        if store_layers:
            out[f'enc_out_{i}'] = t.numpy()
    for i, t in enumerate(dec_outs):
        out[f'dec_stats_{i}'] = stats(t)
        if store_layers:
            out[f'dec_out_{i}'] = t.numpy()
    if store_weights:
        for k, v in state['encoder'].items():
            out['encoder/' + k] = v.numpy()
        for k, v in state['decoder'].items():
            out['decoder/' + k] = v.numpy()
    out['cfg_json'] = np.frombuffer(json.dumps(dict(cfg, seed=seed)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(name, 'y', tuple(y.shape), 'x_r', tuple(x_r[0].shape),
          '|y|max %.3f' % y.abs().max().item(), 'size %.1f KB' % (os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024))


def make_variant_case(ref, name, kw, tile, seed):
    """Variants whose parameters are not produced by synth.synthetic_state (BatchNorm, groups, channel expansion,
    multiscale colour layers): the reference builds the modules under torch.manual_seed(seed), BatchNorm gets
    non-trivial affine parameters and running statistics, and the fixture stores every state-dict tensor next to
    the reference's eval-mode outputs.  (GDN arithmetic is the oracle's, as in make_case.)"""
    torch.manual_seed(seed)
    enc = ref.Analyzer(**kw)
    dec = ref.Synthesizer(**kw)
    g = torch.Generator().manual_seed(seed + 1)
    for mod in list(enc.modules()) + list(dec.modules()):
        if isinstance(mod, nn.BatchNorm2d):
            with torch.no_grad():
                mod.weight.uniform_(0.5, 1.5, generator=g)
                mod.bias.uniform_(-0.2, 0.2, generator=g)
                mod.running_mean.uniform_(-0.2, 0.2, generator=g)
                mod.running_var.uniform_(0.5, 1.5, generator=g)
    enc.eval()
    dec.eval()
    out = {}
    with torch.no_grad():
        x = O.tile_to_input(tile)
        y = enc(x)
        yq = torch.round(y)
        x_r, brg = dec(yq)
    out['tile'] = tile
    out['y'] = y.numpy()
    for i, t in enumerate(x_r):
        if t is not None:
            out[f'x_r_{i}'] = t.numpy()
    out['x_r_none'] = np.array([t is None for t in x_r])
    for i, t in enumerate(brg):
        out[f'brg_stats_{i}'] = stats(t)
        out[f'brg_shape_{i}'] = np.array(t.shape)
    for k, v in enc.state_dict().items():
        out['encoder/' + k] = v.numpy()
    for k, v in dec.state_dict().items():
        out['decoder/' + k] = v.numpy()
    out['cfg_json'] = np.frombuffer(json.dumps(dict(kw, seed=seed)).encode(), dtype=np.uint8)
    np.savez_compressed(os.path.join(GOLD, name + '.npz'), **out)
    print(name, 'y', tuple(y.shape), 'x_r', [None if t is None else tuple(t.shape) for t in x_r],
          'size %.1f KB' % (os.path.getsize(os.path.join(GOLD, name + '.npz')) / 1024))


def make_init_fixture(ref):
    """Pins initialize_weights (_autoencoders.py:37-42) and the state-dict key list."""
    torch.manual_seed(0)
    enc = ref.Analyzer(3, 8, 16, 3, act_layer_type='GDN')
    dec = ref.Synthesizer(3, 8, 16, 3, act_layer_type='GDN')
    out = {}
    for k, v in enc.state_dict().items():
        out['encoder/' + k] = v.numpy()
    for k, v in dec.state_dict().items():
        out['decoder/' + k] = v.numpy()
    np.savez_compressed(os.path.join(GOLD, 'ref_init_seed0.npz'), **out)
    print('ref_init_seed0', list(out.keys()))


def load_reference_criteria():
    """The reference's loss classes (src/models/criteria/_ratedist.py, _lossutils.py, _classification.py) loaded by
    file path as a throw-away package; pytorch_msssim (absent from this image, only used by the MS-SSIM distortions)
    is stubbed exactly as compressai is for _autoencoders.py."""
    base = '/root/reference/src/models/criteria'
    stub = types.ModuleType('pytorch_msssim')
    stub.ms_ssim = None
    sys.modules['pytorch_msssim'] = stub
    pkg = types.ModuleType('refcrit')
    pkg.__path__ = [base]
    sys.modules['refcrit'] = pkg
    mods = {}
    for name in ('_ratedist', '_classification', '_lossutils'):
        spec = importlib.util.spec_from_file_location(f'refcrit.{name}', os.path.join(base, name + '.py'))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[f'refcrit.{name}'] = mod
        spec.loader.exec_module(mod)
        mods[name] = mod
    return mods


def make_loss_case():
    """Values of the reference's training objective (GeneralLoss via setup_loss('RateMSE'), RateLoss, DistMSELoss;
    _lossutils.py:5-109, _ratedist.py:45-63) on seeded tensors -> tests/golden/ref_loss_ratemse.npz.  `entropy_loss`
    is whatever net['fact_ent'].module.loss() returns: a stand-in module hands back a fixed tensor."""
    crit = load_reference_criteria()
    g = torch.Generator().manual_seed(77)
    x = torch.rand(3, 3, 40, 56, generator=g)
    x_r = (x + 0.05 * torch.randn(3, 3, 40, 56, generator=g)).clamp(0, 1)
    p_y = torch.rand(3, 24, 5, 7, generator=g).clamp_min(1e-9)
    aux = torch.tensor(12.375)

    class _Fe:
        def loss(self):
            return aux

    class _Wrapped:
        module = _Fe()

    out = {}
    for lam in (0.01, 0.1):
        crit_obj = crit['_lossutils'].setup_loss('RateMSE', distortion_lambda=lam)
        ld = crit_obj(inputs=x, outputs=dict(x_r=[x_r, None], p_y=p_y, y=None), net={'fact_ent': _Wrapped()})
        tag = str(lam).replace('.', 'p')
        out[f'loss_{tag}'] = np.float64(float(ld['loss']))
        out[f'dist_loss_{tag}'] = np.float64(float(ld['dist_loss']))
        out[f'dist0_{tag}'] = np.float64(float(ld['dist'][0]))
        out[f'rate_loss_{tag}'] = np.float64(float(ld['rate_loss']))
        out[f'entropy_loss_{tag}'] = np.float64(float(ld['entropy_loss']))
        assert sorted(ld.keys()) == ['channel_e', 'dist', 'dist_loss', 'entropy_loss', 'loss', 'rate_loss'], sorted(ld.keys())
    rate = crit['_ratedist'].RateLoss()(x=x, p_y=p_y)['rate_loss']
    mse = crit['_ratedist'].DistMSELoss()(x=x, x_r=[x_r])['dist'][0]
    np.savez_compressed(os.path.join(GOLD, 'ref_loss_ratemse.npz'), x=x.numpy(), x_r=x_r.numpy(), p_y=p_y.numpy(),
                        aux=np.float64(float(aux)), rate_only=np.float64(float(rate)), mse_only=np.float64(float(mse)),
                        keys_json=np.frombuffer(json.dumps(sorted(ld.keys())).encode(), dtype=np.uint8), **out)
    print('ref_loss_ratemse:', {k: float(v) for k, v in out.items()})


def main():
    os.makedirs(GOLD, exist_ok=True)
    ref = load_reference()
    small = dict(synth.CANONICAL, channels_net=8, channels_bn=16, compression_level=3)
    small_noact = dict(small, act_layer_type=None)
    mnist = dict(small, channels_org=1)
    k5 = dict(small, kernel_size=5, bias=True)
    rng = np.random.default_rng(99)

    make_init_fixture(ref)
    # reference-pinned end to end (no GDN): even and ragged sizes
    make_case(ref, 'noact_small_40x56', small_noact, synth.histo_tile(40, 1, 56), 11, True, True)
    make_case(ref, 'noact_small_37x45', small_noact, rng.integers(0, 256, (37, 45, 3), dtype=np.uint8), 12, True, True)
    # GDN variant (conv reference-pinned, GDN restated)
    make_case(ref, 'gdn_small_40x56', small, synth.histo_tile(40, 2, 56), 13, True, True)
    make_case(ref, 'gdn_small_37x45', small, rng.integers(0, 256, (37, 45, 3), dtype=np.uint8), 14, True, True)
    make_case(ref, 'gdn_mnist_32x32', mnist, synth.mnist_like(1)[0], 15, True, True)
    make_case(ref, 'gdn_k5bias_48x48', k5, synth.histo_tile(48, 3), 16, True, True)
    # canonical 128/192/L4 (weights rebuilt from the seed by synth.synthetic_state)
    # LeakyReLU / ReLU units (stride-1 pre-convolutions; reference-pinned end to end, no GDN)
    make_case(ref, 'lrelu_bias_small_40x56', dict(small, act_layer_type='LeakyReLU', bias=True), synth.histo_tile(40, 6, 56), 21, True, True)
    make_case(ref, 'relu_small_37x45', dict(small, act_layer_type='ReLU'), rng.integers(0, 256, (37, 45, 3), dtype=np.uint8), 22, True, True)
    make_case(ref, 'lrelu_k5_mid_48x48', dict(small, act_layer_type='LeakyReLU', kernel_size=5, channels_net=40, channels_bn=24), synth.histo_tile(48, 7), 23, False, False)
    # variants folded on upload: BatchNorm (eval), grouped (depthwise) layers, channel expansion
    vk = dict(channels_org=3, channels_net=8, channels_bn=16, compression_level=3, channels_expansion=1, kernel_size=3,
              groups=False, batch_norm=False, dropout=0.0, bias=False, use_residual=False, act_layer_type=None)
    make_variant_case(ref, 'var_bn_gdn_40x56', dict(vk, batch_norm=True, act_layer_type='GDN'),
                      synth.histo_tile(40, 8, 56), 31)
    make_variant_case(ref, 'var_bn_lrelu_bias_37x45', dict(vk, batch_norm=True, bias=True, act_layer_type='LeakyReLU',
                                                           dropout=0.25),
                      np.random.default_rng(199).integers(0, 256, (37, 45, 3), dtype=np.uint8), 32)
    make_variant_case(ref, 'var_expansion2_gdn_48x48', dict(vk, channels_net=4, channels_bn=24, channels_expansion=2,
                                                            act_layer_type='GDN'), synth.histo_tile(48, 9), 33)
    make_variant_case(ref, 'var_groups_relu_40x40', dict(vk, channels_net=3, channels_bn=3, groups=True, bias=True,
                                                         act_layer_type='ReLU'), synth.histo_tile(40, 10), 34)
    make_variant_case(ref, 'var_groups_k5_32x48', dict(vk, channels_net=3, channels_bn=3, groups=True, kernel_size=5),
                      synth.histo_tile(32, 11, 48), 35)
    make_variant_case(ref, 'var_multiscale_lrelu_bias_40x56', dict(vk, bias=True, act_layer_type='LeakyReLU',
                                                                   multiscale_analysis=True),
                      synth.histo_tile(40, 12, 56), 36)
    make_variant_case(ref, 'var_multiscale_gdn_k5_48x48', dict(vk, kernel_size=5, act_layer_type='GDN',
                                                               multiscale_analysis=True), synth.histo_tile(48, 13), 37)
    # residual units (res_model + unit input, then the strided layer)
    make_variant_case(ref, 'var_res_gdn_40x56', dict(vk, use_residual=True, act_layer_type='GDN'),
                      synth.histo_tile(40, 14, 56), 38)
    make_variant_case(ref, 'var_res_lrelu_bn_bias_37x45', dict(vk, use_residual=True, batch_norm=True, bias=True,
                                                               act_layer_type='LeakyReLU'),
                      np.random.default_rng(299).integers(0, 256, (37, 45, 3), dtype=np.uint8), 39)
    make_variant_case(ref, 'var_res_none_k5_48x48', dict(vk, use_residual=True, kernel_size=5), synth.histo_tile(48, 15), 40)
    make_variant_case(ref, 'var_res_relu_mid_32x32', dict(vk, use_residual=True, channels_net=40, channels_bn=24,
                                                          act_layer_type='ReLU'), synth.histo_tile(32, 16), 41)
    make_case(ref, 'gdn_canonical_64x64', synth.CANONICAL, synth.histo_tile(64, 4), 17, False, False)
    make_case(ref, 'gdn_canonical_96x80', synth.CANONICAL, rng.integers(0, 256, (96, 80, 3), dtype=np.uint8), 18, False, False)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'loss':
        make_loss_case()  # (only the loss fixture; the model fixtures are unchanged)
    else:
        main()
        make_loss_case()
