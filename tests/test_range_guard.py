"""f16x3 range guard (include/cae_hip.h, "VALID RANGE of f16x3"), on the GPU.

The default arithmetic splits every fp32 operand into two f16 halves; f16 carries 5 exponent bits.  The reference
computes in fp32 throughout (_autoencoders.py:78-85, :204-211), so magnitudes a trained model may produce must give the
reference's results: GDN / IGDN squares of any finite magnitude stay on the f16x3 kernels (per-pixel power-of-two
scale), and a value that cannot be stored (|v| > 65504) makes the call repeat on the exact-fp32 kernels.
Tolerance: the usual 1e-4, relative to the tensor's largest magnitude (the tensors here reach 1e5 .. 1e6).
"""
import ctypes
import struct

import numpy as np
import pytest
import torch

from conftest import oracle_layers

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope='module')
def cae(built_lib):
    import cnn_autoencoder_amd as cae
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return cae


def _model(cae, state, precision='f16x3'):
    model = cae.autoencoder_from_state_dict(state)
    for k in ('encoder', 'decoder'):
        model[k].module.precision = precision
    return model


def _close(got, want, what):
    scale = max(1.0, float(want.abs().max()))
    err = float((got - want).abs().max()) / scale
    assert err < RTOL, f'{what}: max error {err:.3e} of the largest magnitude {scale:.3e}'


def _track_maxima(O, x, layers, synthesis=False):
    """largest |value| before each GDN / IGDN and of each unit's output (CPU oracle)"""
    pre, post, cur = [], [], x
    for l in layers:
        y = (O.deconv_s2 if synthesis else O.reflect_conv_s2)(cur, l['weight'], l.get('bias'))
        pre.append(float(y.abs().max()))
        if l.get('beta') is not None:
            y = O.gdn_forward(y, l['beta'], l['gamma'], inverse=synthesis)
        post.append(float(y.abs().max()))
        cur = y
    return pre, post


def test_gdn_squares_of_large_activations_stay_on_f16x3(cae):
    """pre-GDN |y| of 1e3 .. 3e4 (the unscaled squares overflowed f16 beyond 255.9) and pre-IGDN |y| of several
    hundred: no fallback is needed, the f16x3 kernels themselves match the oracle."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    state = synth.synthetic_state(cfg, seed=11)
    state['encoder']['analysis_track.0.model.0.weight'] *= 3.0e4   # GDN brings the magnitudes back to O(1 / sqrt(gamma))
    state['encoder']['analysis_track.1.model.0.weight'] *= 4.0e2
    for i in range(2):  # a weak IGDN: its output grows like gamma |y|^2
        g = state['decoder'][f'synthesis_track.{i}.model.1.gamma']
        state['decoder'][f'synthesis_track.{i}.model.1.gamma'] = g * 0.1
    state['decoder']['synthesis_track.0.model.0.weight'] *= 100.0
    state['decoder']['synthesis_track.1.model.0.weight'] *= 0.05
    model = _model(cae, state)
    enc, dec = model['encoder'].module, model['decoder'].module
    tiles = synth.uniform_tiles(2, 72, 88)
    x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
    enc_l, dec_l = oracle_layers(state, 'encoder'), oracle_layers(state, 'decoder')
    pre, post = _track_maxima(O, x, enc_l)
    assert pre[0] > 1e4 and pre[1] > 1e3 and max(post[:-1]) < 6e4, (pre, post)  # the premise of this test
    y_ref, _ = O.analysis_forward(x, enc_l)
    y = enc.forward_u8(torch.from_numpy(tiles).cuda()).cpu()
    _close(y, y_ref, 'latents')
    assert enc.fp32_fallbacks == 0

    yq = torch.round(y_ref).clamp(-40, 40)
    pre, post = _track_maxima(O, yq, dec_l, synthesis=True)
    assert max(pre[:2]) > 300 and max(post[:-1]) < 6e4, (pre, post)
    x_ref, _ = O.synthesis_forward(yq, dec_l)
    x_r, brg = dec(yq.cuda())
    _close(x_r[0].cpu(), x_ref, 'reconstruction')
    assert dec.fp32_fallbacks == 0


def test_unstorable_activations_fall_back_to_fp32(cae):
    """activations of 1e5 .. 1e6 between layers (no GDN to bring them back) and latents of 1e3 into an IGDN stack: the
    split format cannot hold them, the calls are repeated on the fp32 kernels and match the oracle."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3, act_layer_type=None)
    state = synth.synthetic_state(cfg, seed=12)
    state['encoder']['analysis_track.0.model.0.weight'] *= 3.0e5
    state['encoder']['analysis_track.2.model.0.weight'] *= 1.0e-4
    model = _model(cae, state)
    enc = model['encoder'].module
    tiles = synth.uniform_tiles(2, 64, 80)
    x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
    enc_l = oracle_layers(state, 'encoder')
    pre, _ = _track_maxima(O, x, enc_l)
    assert pre[0] > 1e5, pre
    y_ref, _ = O.analysis_forward(x, enc_l)
    y = enc.forward_u8(torch.from_numpy(tiles).cuda()).cpu()
    _close(y, y_ref, 'latents')
    assert enc.fp32_fallbacks == 1
    y2 = enc(x.cuda()).cpu()  # float entry point: same guard
    _close(y2, y_ref, 'latents (float input)')
    assert enc.fp32_fallbacks == 2

    gcfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    gstate = synth.synthetic_state(gcfg, seed=13)
    gmodel = _model(cae, gstate)
    dec = gmodel['decoder'].module
    yq = torch.round(torch.randn(2, 48, 5, 7) * 1.0e3)
    dec_l = oracle_layers(gstate, 'decoder')
    pre, post = _track_maxima(O, yq, dec_l, synthesis=True)
    assert max(post[:-1]) > 1e5, post
    x_ref, _ = O.synthesis_forward(yq, dec_l)
    x_r, brg = dec(yq.cuda())
    _close(x_r[0].cpu(), x_ref, 'reconstruction')
    assert dec.fp32_fallbacks == 1
    u8 = dec.forward_u8(yq.cuda()).cpu()
    own = (x_r[0].cpu() * 255.0).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    assert torch.equal(u8, own)

    # a float input beyond the f16 range (and a NaN) is caught at the layout conversion
    xb = x.clone()
    xb[0, 0, 3, 4] = 7.0e4
    yb_ref, _ = O.analysis_forward(xb, enc_l)
    before = enc.fp32_fallbacks
    _close(enc(xb.cuda()).cpu(), yb_ref, 'latents (input beyond the f16 range)')
    assert enc.fp32_fallbacks == before + 1


def test_weights_beyond_f16_run_on_fp32(cae):
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import _lib, synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=2)
    state = synth.synthetic_state(cfg, seed=14)
    state['encoder']['analysis_track.0.model.0.weight'][3, 1, 1, 1] = 1.0e5
    model = _model(cae, state)
    enc = model['encoder'].module
    tiles = synth.uniform_tiles(1, 40, 48)
    x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
    y_ref, _ = O.analysis_forward(x, oracle_layers(state, 'encoder'))
    _close(enc.forward_u8(torch.from_numpy(tiles).cuda()).cpu(), y_ref, 'latents')
    prec = ctypes.c_int(-1)
    _lib.check(_lib.lib().cae_model_effective_precision(enc._sync().ptr, ctypes.byref(prec)))
    assert prec.value == 0 and enc.precision_code() == 1 and enc.fp32_fallbacks == 0


def test_stress_state_through_the_codec(cae):
    """SURVEY 8d `stress` variant (last analysis layer x40: latents leave the CDF support, bypass coding) through
    codec.encode / decode on the default arithmetic, against the oracle codec."""
    from oracle import c_oracle as C
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48)
    state = synth.synthetic_state(cfg, seed=3, stress=True)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    assert codec._model['encoder'].module.precision_code() == 1
    o = O.EntropyBottleneckOracle(48)
    o.load(state['fact_ent'])
    o.update()
    enc_l, dec_l = oracle_layers(state, 'encoder'), oracle_layers(state, 'decoder')
    for tile in (synth.histo_tile(96, 4, 80), synth.uniform_tiles(1, 64, 64)[0]):
        buf = codec.encode(tile)
        assert struct.unpack('>QQ', buf[:16]) == tile.shape[:2]
        y_gpu = codec._model['encoder'].module.forward_u8(torch.from_numpy(tile)[None].cuda()).cpu()
        y_ref, _ = O.analysis_forward(O.tile_to_input(tile), enc_l)
        _close(y_gpu, y_ref, 'latents')
        sym = o.symbols(y_gpu)
        assert int(sym.abs().max()) > 30  # outside the support of the initial CDF tables: bypass symbols
        assert buf[16:] == o.compress(y_gpu, C.rans_encode_with_indexes)[0]
        flips = int((sym != o.symbols(y_ref)).sum())
        assert flips <= max(1, sym.numel() // 1000), f'{flips} symbol flips vs the oracle latents'
        rec = codec.decode(buf)
        ref_rec = O.codec_decode(buf, dec_l, o, C.rans_decode_with_indexes)
        diff = np.abs(rec.astype(int) - ref_rec.astype(int))
        assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


def test_pipelined_drivers_repeat_only_the_batches_that_overflow(cae):
    """Bright tiles overflow the f16 range in this (activation-free) model, dark ones do not: the pipelined round trip
    and the one-way streams repeat exactly those batches on fp32 and give the results of an all-fp32 run."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import slide, synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3, act_layer_type=None)
    state = synth.synthetic_state(cfg, seed=15)
    state['encoder']['analysis_track.0.model.0.weight'] *= 5.0e4
    state['encoder']['analysis_track.2.model.0.weight'] *= 2.0e-5
    rng = np.random.default_rng(5)
    dark = [rng.integers(0, 12, (3, 64, 96, 3), dtype=np.uint8) for _ in range(3)]
    bright = [rng.integers(200, 256, (3, 64, 96, 3), dtype=np.uint8) for _ in range(2)]
    enc_l = oracle_layers(state, 'encoder')
    to_x = lambda t: torch.from_numpy(t).permute(0, 3, 1, 2).float() / 255.0
    assert max(_track_maxima(O, to_x(dark[0]), enc_l)[1][:-1]) < 5e4
    assert max(_track_maxima(O, to_x(bright[0]), enc_l)[1][:-1]) > 8e4
    order = [dark[0], bright[0], dark[1], dark[2], bright[1]]
    batches = [torch.from_numpy(b).cuda() for b in order]

    ref_codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    for k in ('encoder', 'decoder'):
        ref_codec._model[k].module.precision = 'fp32'
    ref = slide.SlideCoder(ref_codec)
    want = [ref.roundtrip(b) for b in batches]

    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    coder = slide.SlideCoder(codec)
    coder.depth = 2
    enc = codec._model['encoder'].module
    assert enc.precision_code() == 1
    stats, payloads = coder.run(batches, keep_payloads=True)
    assert enc.fp32_fallbacks == 2
    # dark batches ran on f16x3 (same symbols up to float-noise flips are not guaranteed bit for bit against fp32):
    # the bright ones must equal the fp32 run exactly, the dark ones within the flip allowance
    for k, (pl, (pl_ref, _, st_ref)) in enumerate(zip(payloads, want)):
        if k in (1, 4):
            assert pl == pl_ref
            assert torch.equal(stats[3 * k:3 * k + 3], st_ref)
        else:
            assert all(abs(len(a) - len(b)) <= 8 for a, b in zip(pl, pl_ref))
    got = list(coder.compress_batches(iter(order)))
    assert enc.fp32_fallbacks == 4
    assert got[1] == want[1][0] and got[4] == want[4][0]
    rec = list(coder.decompress_batches(iter([w[0] for w in want]), 64, 96))
    assert len(rec) == 5 and all(int((r.int() - w[1].int()).abs().max()) <= 1 for r, w in zip(rec, want))
