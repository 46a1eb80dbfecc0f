"""GPU parity tests (run on the MI355X box with -m gpu): HIP path through the C ABI vs the CPU oracle
and the golden fixtures emitted by the reference's own classes.

Tolerances: the conv / GDN path is fp32 MFMA (fmaf-chain numerics); north_star asks for reconstructed
pixels within 1e-4 relative fp32, so float tensors are compared with atol = rtol = 1e-4 against the
reference-generated goldens (observed error is ~1e-6).  Integer work (symbols, bitstreams, uint8
tiles given identical inputs) is compared bit-exactly.
"""
import struct

import numpy as np
import pytest
import torch

from conftest import golden_state, load_golden, oracle_layers

pytestmark = pytest.mark.gpu

RTOL = ATOL = 1e-4

CASES = ['noact_small_40x56', 'noact_small_37x45', 'gdn_small_40x56', 'gdn_small_37x45', 'gdn_mnist_32x32',
         'gdn_k5bias_48x48', 'gdn_canonical_64x64', 'gdn_canonical_96x80', 'lrelu_bias_small_40x56',
         'relu_small_37x45', 'lrelu_k5_mid_48x48']


@pytest.fixture(params=['fp32', 'f16x3'], autouse=True)
def precision(request, monkeypatch):
    """Every parity test runs on both arithmetic paths of the conv / GDN contraction: exact fp32 MFMA
    and the f16x3 split (three f16 MFMAs per product, fp32 accumulate) -- same tolerances."""
    monkeypatch.setenv('CAE_PRECISION', request.param)
    return request.param


@pytest.fixture(scope='module')
def cae(built_lib):
    import cnn_autoencoder_amd as cae
    assert torch.cuda.is_available(), 'GPU tests need a HIP device'
    return cae


def build_model(cae, state):
    return cae.autoencoder_from_state_dict(state)


@pytest.mark.parametrize('name', CASES)
def test_analysis_matches_reference_golden(cae, name):
    g, cfg = load_golden(name)
    state = golden_state(g, cfg)
    model = build_model(cae, state)
    tile = g['tile']
    x = torch.from_numpy(tile).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    y = model['encoder'](x.cuda()).cpu().numpy()
    assert y.shape == g['y'].shape
    np.testing.assert_allclose(y, g['y'], rtol=RTOL, atol=ATOL)
    # uint8 side door (fused /255) must agree with the float entry point bit for bit
    y8 = model['encoder'].module.forward_u8(torch.from_numpy(tile)[None].cuda()).cpu().numpy()
    assert np.array_equal(y8, y)


@pytest.mark.parametrize('name', CASES)
def test_synthesis_matches_reference_golden(cae, name):
    g, cfg = load_golden(name)
    state = golden_state(g, cfg)
    model = build_model(cae, state)
    yq = torch.round(torch.from_numpy(g['y']))
    x_r, brg = model['decoder'](yq.cuda())
    L = cfg['compression_level']
    assert len(x_r) == L and all(t is None for t in x_r[1:]) and len(brg) == L
    out = x_r[0].cpu().numpy()
    assert out.shape == g['x_r'].shape
    np.testing.assert_allclose(out, g['x_r'], rtol=RTOL, atol=ATOL)
    if 'dec_out_0' in g.files:
        for i in range(L):
            np.testing.assert_allclose(brg[i].cpu().numpy(), g[f'dec_out_{i}'], rtol=RTOL, atol=ATOL)
    # uint8 epilogue: identical to truncating the GPU's own float output; vs the reference's bytes
    # a 1-LSB flip is possible only where x*255 sits within float noise of an integer
    u8 = model['decoder'].module.forward_u8(yq.cuda()).cpu().numpy()[0]
    own = (x_r[0][0].cpu() * 255.0).clip(0, 255).to(torch.uint8).permute(1, 2, 0).numpy()
    assert np.array_equal(u8, own)
    diff = np.abs(u8.astype(int) - g['x_r_u8'].astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3


@pytest.mark.parametrize('name', ['var_bn_gdn_40x56', 'var_bn_lrelu_bias_37x45', 'var_expansion2_gdn_48x48',
                                  'var_groups_relu_40x40', 'var_groups_k5_32x48', 'var_multiscale_lrelu_bias_40x56',
                                  'var_multiscale_gdn_k5_48x48', 'var_res_gdn_40x56', 'var_res_lrelu_bn_bias_37x45',
                                  'var_res_none_k5_48x48', 'var_res_relu_mid_32x32'])
def test_variant_goldens(cae, name):
    """BatchNorm (eval), grouped layers and channel expansion against the reference's own outputs."""
    from test_host import variant_modules
    g, cfg = load_golden(name)
    enc, dec = variant_modules(cae, g, cfg)
    x = torch.from_numpy(g['tile']).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    y = enc.cuda()(x.cuda()).cpu().numpy()
    np.testing.assert_allclose(y, g['y'], rtol=RTOL, atol=ATOL)
    x_r, brg = dec.cuda()(torch.round(torch.from_numpy(g['y'])).cuda())
    # tolerance stated: 1e-4 relative, with the absolute floor scaled to the tensor (untrained residual / IGDN stacks
    # produce |x_r| up to ~20, where fp32 summation-order noise alone is ~1e-4 on elements that cancel to ~0)
    scale = max(1.0, float(np.abs(g['x_r_0']).max()))
    np.testing.assert_allclose(x_r[0].cpu().numpy(), g['x_r_0'], rtol=RTOL, atol=ATOL * scale)
    for j in range(1, len(x_r)):  # multiscale colour layers, or None as the reference returns
        if g['x_r_none'][j]:
            assert x_r[j] is None
        else:
            np.testing.assert_allclose(x_r[j].cpu().numpy(), g[f'x_r_{j}'], rtol=RTOL, atol=ATOL)
    assert [tuple(t.shape) for t in brg] == [tuple(g[f'brg_shape_{i}']) for i in range(len(brg))]
    for i, t in enumerate(brg):
        st = np.array([t.double().sum().item(), t.double().abs().sum().item(), (t.double() ** 2).sum().item()])
        np.testing.assert_allclose(st, g[f'brg_stats_{i}'], rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('inverse', [False, True])
@pytest.mark.parametrize('channels', [16, 128, 192])
def test_gdn_layer(cae, inverse, channels):
    from oracle import cae_oracle as O
    torch.manual_seed(channels + inverse)
    cfg = dict(channels_org=3, channels_net=channels, channels_bn=channels, compression_level=2, act_layer_type='GDN')
    mod = (cae.Synthesizer if inverse else cae.Analyzer)(**cfg)
    track = mod.synthesis_track if inverse else mod.analysis_track
    gdn = track[0].model[1]
    with torch.no_grad():
        gdn.beta.copy_(torch.sqrt(torch.rand(channels) + 0.5))
        gdn.gamma.copy_(torch.sqrt(0.1 * torch.eye(channels) + 0.02 * torch.rand(channels, channels)))
    x = torch.randn(2, channels, 9, 13)
    ref = O.gdn_forward(x, gdn.beta.detach(), gdn.gamma.detach(), inverse=inverse)
    out = gdn(x.cuda()).cpu()
    np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=RTOL, atol=ATOL)


def _eb_pair(cae, channels, fit, seed=0):
    from oracle import cae_oracle as O
    torch.manual_seed(seed)
    eb = cae.EntropyBottleneck(channels, filters=[3] * 4).eval()
    if fit:
        eb.fit_quantiles()
    eb.update(force=True)
    o = O.EntropyBottleneckOracle(channels)
    o.load(eb.state_dict())
    o.update()
    assert torch.equal(o._quantized_cdf, eb._quantized_cdf.cpu())
    return eb.cuda(), o


@pytest.mark.parametrize('fit,scale', [(False, 6.0), (True, 6.0), (False, 300.0), (True, 2000.0)])
def test_quantize_and_bitstream_bit_exact(cae, fit, scale):
    """Integer step: symbols and rANS bytes identical to the oracle's for identical latents
    (incl. bypass-heavy inputs), and decompress inverts compress exactly."""
    from oracle import c_oracle as C
    eb, o = _eb_pair(cae, 24, fit)
    torch.manual_seed(3)
    y = torch.randn(3, 24, 7, 5) * scale
    y[0, 0, 0, :4] = torch.tensor([0.5, 1.5, 2.5, -0.5])  # half-to-even ties
    sym = eb.quantize_symbols(y.cuda()).cpu()
    assert torch.equal(sym, o.symbols(y))
    strings = eb.compress(y.cuda())
    ref = o.compress(y, C.rans_encode_with_indexes)
    assert strings == ref
    yq = eb.decompress(strings, (7, 5)).cpu()
    yq_ref, _ = o.forward(y)
    assert torch.equal(yq, yq_ref)
    assert torch.equal(o.decompress(strings, (7, 5), C.rans_decode_with_indexes), yq_ref)


@pytest.mark.parametrize('form', ['plain', 'sign_trick'])
@pytest.mark.parametrize('filters,fit,scale', [((3, 3, 3, 3), True, 6.0), ((3, 3, 3, 3), False, 40.0),
                                               ((5, 5), True, 6.0), ((2, 4, 3), True, 3.0), ((1,), False, 3.0)])
def test_likelihood_matches_oracle(cae, filters, fit, scale, form):
    """EntropyBottleneck.__call__ (eval): y_hat exact, likelihood within 1e-4 relative (fp32 transcendental
    functions differ by ulps between torch-CPU and the device), rate estimate within 1e-5; both floating-point
    forms of the bin probability (entropy.LIKELIHOOD_FORMS).  The plain form sigmoid(u) - sigmoid(l) cancels near
    sigmoid = 1: its absolute noise is one ulp of 1 (6e-8), stated as the absolute floor."""
    from oracle import cae_oracle as O
    torch.manual_seed(5)
    atol = 2e-7 if form == 'plain' else 1e-12
    eb = cae.EntropyBottleneck(20, filters=filters, likelihood_form=form).eval()
    with torch.no_grad():  # non-trivial factors and matrices (init has factor = 0)
        for n, p in eb.named_parameters():
            if n.startswith('_factor'):
                p.uniform_(-1.0, 1.0)
            elif n.startswith('_matrix'):
                p.add_(torch.randn_like(p) * 0.3)
    if fit:
        eb.fit_quantiles()
    eb.update(force=True)
    o = O.EntropyBottleneckOracle(20, filters=filters, likelihood_form=form)
    o.load(eb.state_dict())
    eb = eb.cuda()
    y = torch.randn(3, 20, 9, 13) * scale
    y_ref, p_ref = o.forward(y)
    with torch.no_grad():
        y_hat, p = eb(y.cuda())
    assert torch.equal(y_hat.cpu(), y_ref)
    # tolerance stated: 1e-4 relative on the likelihood, floor 1e-12 absolute (values are >= the 1e-9 bound)
    np.testing.assert_allclose(p.cpu().numpy(), p_ref.numpy(), rtol=1e-4, atol=atol)
    assert float(p.min()) >= 1e-9 * (1 - 1e-6)
    bits = eb.rate_bits(y.cuda()).cpu().numpy()
    bits_ref = -torch.log2(p_ref.double()).sum(dim=(1, 2, 3)).numpy()
    if form == 'sign_trick':  # (plain: tail probabilities of ~1e-7 carry the cancellation noise into log2)
        np.testing.assert_allclose(bits, bits_ref, rtol=1e-5)
    else:
        np.testing.assert_allclose(bits, -torch.log2(p.double().cpu()).sum(dim=(1, 2, 3)).numpy(), rtol=1e-5)
    # the HIP path and the autograd (torch-op) path of the same module agree
    y_t, p_t = eb(y.cuda().requires_grad_(True))
    assert torch.equal(y_t.detach(), y_hat)
    np.testing.assert_allclose(p_t.detach().cpu().numpy(), p.cpu().numpy(), rtol=1e-4, atol=atol)


def test_rate_estimate_tracks_coded_size(cae):
    """-sum log2 p (GPU) against the real rANS payload: the estimate is the coder's ideal length, so the coded
    size must sit within a few percent above/below it on in-support latents."""
    eb, _ = _eb_pair(cae, 48, True)
    torch.manual_seed(7)
    y = (torch.randn(4, 48, 32, 32) * 4.0).cuda()
    bits = eb.rate_bits(y).cpu().numpy()
    coded = np.array([8 * len(s) for s in eb.compress(y)], dtype=np.float64)
    assert np.all(np.abs(coded - bits) / bits < 0.03)


@pytest.mark.parametrize('name', ['gdn_small_40x56', 'gdn_canonical_64x64', 'gdn_mnist_32x32'])
def test_codec_encode_decode(cae, name):
    """Codec byte format + end-to-end: header, payload = oracle coder on the GPU's latents,
    decode(encode(tile)) equals the oracle's decode of the same bytes up to float-noise LSB flips."""
    from oracle import c_oracle as C
    from oracle import cae_oracle as O
    g, cfg = load_golden(name)
    state = golden_state(g, cfg)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    tile = g['tile']
    buf = codec.encode(tile)
    h, w = struct.unpack('>QQ', buf[:16])
    assert (h, w) == tile.shape[:2]
    # the oracle entropy model built from the same parameters
    o = O.EntropyBottleneckOracle(cfg['channels_bn'])
    o.load(state['fact_ent'])
    o.update()
    y_gpu = codec._model['encoder'].module.forward_u8(torch.from_numpy(tile)[None].cuda()).cpu()
    assert buf[16:] == o.compress(y_gpu, C.rans_encode_with_indexes)[0]
    # symbol agreement with the reference-generated latents (float cliff: report flips, expect none here)
    flips = (o.symbols(y_gpu) != o.symbols(torch.from_numpy(g['y']))).sum().item()
    assert flips <= max(1, y_gpu.numel() // 1000), f'{flips} symbol flips vs reference latents'
    rec = codec.decode(buf)
    assert rec.dtype == np.uint8 and rec.shape == g['x_r_u8'].shape
    ref_rec = O.codec_decode(buf, oracle_layers(state, 'decoder'), o, C.rans_decode_with_indexes)
    diff = np.abs(rec.astype(int) - ref_rec.astype(int))
    assert diff.max() <= 1 and (diff > 0).mean() < 1e-3
    out = np.empty_like(rec)
    assert codec.decode(buf, out=out) is not None and np.array_equal(out, rec)
    # numcodecs config round trip (Codec.from_config(cls(**config)))
    cfgd = codec.get_config()
    assert cfgd['id'] == 'cae' and set(cfgd) == {'id', 'checkpoint', 'gpu'}


def test_codec_bottleneck(cae):
    from oracle import c_oracle as C
    from oracle import cae_oracle as O
    torch.manual_seed(5)
    eb = cae.EntropyBottleneck(16, filters=[3] * 4)
    codec = cae.ConvolutionalAutoencoderBottleneck(channels_bn=16, fact_ent=eb)
    cfgd = codec.get_config()
    assert set(cfgd) == {'id', 'filters', 'channels_bn', 'fact_ent_checkpoint'} and cfgd['id'] == 'cae_bn'
    clone = cae.ConvolutionalAutoencoderBottleneck.from_config(cfgd)
    lat = (np.random.default_rng(0).standard_normal((6, 9, 16)) * 4).astype(np.float32)
    buf = codec.encode(lat)
    assert struct.unpack('>QQ', buf[:16]) == (6, 9)
    assert clone.encode(lat) == buf
    o = O.EntropyBottleneckOracle(16)
    o.load(eb.state_dict())
    o.update()
    y = torch.from_numpy(lat).permute(2, 0, 1)[None]
    assert buf[16:] == o.compress(y, C.rans_encode_with_indexes)[0]
    dec = codec.decode(buf)
    assert dec.shape == (6, 9, 16) and dec.dtype == np.float32
    assert np.array_equal(dec, o.forward(y)[0][0].permute(1, 2, 0).numpy())


def test_batch_equals_single(cae):
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=2)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    tiles = synth.uniform_tiles(5, 64, 96)
    batch = codec.encode_batch(tiles)
    assert batch == [codec.encode(t) for t in tiles]
    rec = codec.decode_batch(batch)
    assert all(np.array_equal(rec[i], codec.decode(batch[i])) for i in range(5))


def test_full_size_properties(cae):
    """BASELINE cfg at full tile size (canonical 128/192/L4, 1024x1024x3): size-independent properties."""
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(synth.CANONICAL, seed=0)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    eb = codec._model['fact_ent'].module
    tiles = np.stack([synth.histo_tile(1024, 0), synth.uniform_tiles(1, 1024)[0]])
    bufs = codec.encode_batch(tiles)
    assert len(bufs) == 2 and all(struct.unpack('>QQ', b[:16]) == (1024, 1024) for b in bufs)
    # decode -> symbols -> re-encode is the identity on bitstreams (coder round trip at full size)
    sym = eb.decode_symbols([b[16:] for b in bufs], 64 * 64)
    assert [b[16:] for b in bufs] == eb.encode_symbols(sym)
    # analysis is deterministic and batch-invariant
    y2 = codec._model['encoder'].module.forward_u8(torch.from_numpy(tiles).cuda())
    y1 = codec._model['encoder'].module.forward_u8(torch.from_numpy(tiles[1:]).cuda())
    assert torch.equal(y2[1:], y1)
    assert torch.isfinite(y2).all()
    rec = codec.decode_batch(bufs)
    assert rec.shape == (2, 1024, 1024, 3) and rec.dtype == np.uint8
    # shift-equivariance away from borders: a tile shifted by 16 px gives latents shifted by 1
    big = synth.histo_tile(1024 + 16, 5)
    ya = codec._model['encoder'].module.forward_u8(torch.from_numpy(big[:1024, :1024][None].copy()).cuda())
    yb = codec._model['encoder'].module.forward_u8(torch.from_numpy(big[16:, 16:][None].copy()).cuda())
    np.testing.assert_allclose(ya[0, :, 3:-2, 3:-2].cpu().numpy(), yb[0, :, 2:-3, 2:-3].cpu().numpy(),
                               rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize('name', ['lrelu_bias_small_40x56', 'relu_small_37x45', 'lrelu_k5_mid_48x48'])
def test_activation_units_stay_on_the_selected_arithmetic(cae, name, precision):
    """LeakyReLU / ReLU units (the reference's default act_layer_type; stride-1 pre-convolutions, _autoencoders.py:62-76,
    :187-202) run on the split-f16 kernels when f16x3 is selected (round 2 sent them to the fp32 kernels silently): the
    handle reports the arithmetic it uses, no fp32 repeat happens, and the codec round trip matches the oracle."""
    import ctypes
    from cnn_autoencoder_amd import _lib
    g, cfg = load_golden(name)
    state = golden_state(g, cfg)
    model = build_model(cae, state)
    enc, dec = model['encoder'].module, model['decoder'].module
    want = 1 if precision == 'f16x3' else 0
    assert enc.precision_code() == want and dec.precision_code() == want
    x = torch.from_numpy(g['tile']).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    y = model['encoder'](x.cuda()).cpu()
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=RTOL, atol=ATOL)
    x_r, _ = model['decoder'](torch.round(torch.from_numpy(g['y'])).cuda())
    np.testing.assert_allclose(x_r[0].cpu().numpy(), g['x_r'], rtol=RTOL, atol=ATOL)
    for track in (enc, dec):
        eff = ctypes.c_int(-1)
        _lib.check(_lib.lib().cae_model_effective_precision(track._sync().ptr, ctypes.byref(eff)))
        assert eff.value == want and track.fp32_fallbacks == 0


@pytest.mark.parametrize('name', ['var_res_gdn_40x56', 'var_res_lrelu_bn_bias_37x45', 'var_res_none_k5_48x48',
                                  'var_res_relu_mid_32x32', 'var_multiscale_lrelu_bias_40x56', 'var_multiscale_gdn_k5_48x48'])
def test_residual_and_multiscale_units_stay_on_the_selected_arithmetic(cae, name, precision):
    """Residual units (stride-1 stages with an (I)GDN / activation + residual-sum epilogue, _autoencoders.py:104-174,
    :230-304) and the multiscale colour layers (:417-436) run on the split-f16 stride-1 kernel when f16x3 is selected
    (rounds 1-2 sent them to the fp32 kernels): the handle reports the arithmetic in use and no fp32 repeat happens.
    (Their outputs against the reference's are test_variant_goldens, which runs under both precisions.)"""
    import ctypes
    from test_host import variant_modules
    from cnn_autoencoder_amd import _lib
    g, cfg = load_golden(name)
    enc, dec = variant_modules(cae, g, cfg)
    want = 1 if precision == 'f16x3' else 0
    assert enc.precision_code() == want and dec.precision_code() == want
    x = torch.from_numpy(g['tile']).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    y = enc.cuda()(x.cuda()).cpu().numpy()
    np.testing.assert_allclose(y, g['y'], rtol=RTOL, atol=ATOL)
    dec.cuda()(torch.round(torch.from_numpy(g['y'])).cuda())
    for track in (enc, dec):
        eff = ctypes.c_int(-1)
        _lib.check(_lib.lib().cae_model_effective_precision(track._sync().ptr, ctypes.byref(eff)))
        assert eff.value == want and track.fp32_fallbacks == 0


@pytest.mark.parametrize('kind', ['histo', 'uniform'])
def test_full_size_tile_against_the_oracle(cae, kind, precision):
    """BASELINE's tile size (canonical 128/192/L4 model, one 1024x1024x3 tile of each synthetic kind), both arithmetic
    paths, against the CPU oracle: latents within 1e-4, payload == the oracle's C coder on the GPU's latents, decode
    of those bytes within 1 LSB of the oracle's decode, and the float -> integer cliff COUNTED: symbols of the GPU
    latents vs symbols of the oracle's own latents (786 432 per tile; a flip needs both latents within float noise of
    the rounding boundary)."""
    from oracle import c_oracle as C
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(synth.CANONICAL, seed=0)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    enc = codec._model['encoder'].module
    assert enc.precision_code() == (1 if precision == 'f16x3' else 0)
    tile = synth.histo_tile(1024, 3) if kind == 'histo' else synth.uniform_tiles(1, 1024)[0]
    o = O.EntropyBottleneckOracle(192)
    o.load(state['fact_ent'])
    o.update()
    y_ref, _ = O.analysis_forward(O.tile_to_input(tile), oracle_layers(state, 'encoder'))
    y_gpu = enc.forward_u8(torch.from_numpy(tile)[None].cuda()).cpu()
    np.testing.assert_allclose(y_gpu.numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)
    buf = codec.encode(tile)
    assert struct.unpack('>QQ', buf[:16]) == (1024, 1024)
    assert buf[16:] == o.compress(y_gpu, C.rans_encode_with_indexes)[0]
    s_gpu, s_ref = o.symbols(y_gpu), o.symbols(y_ref)
    assert s_gpu.numel() == 786432
    flips = s_gpu != s_ref
    n_flips = int(flips.sum())
    assert n_flips <= 16, f'{n_flips} of 786432 symbols differ from the oracle latents\' symbols'
    if n_flips:
        med = o.medians().reshape(1, -1, 1, 1)
        bound = torch.minimum(s_gpu, s_ref).float() + 0.5
        dist = torch.maximum((y_gpu - med - bound).abs(), (y_ref - med - bound).abs())[flips]
        assert int((s_gpu - s_ref).abs().max()) == 1 and float(dist.max()) < 2e-5, float(dist.max())
    rec = codec.decode(buf)
    ref_rec = O.codec_decode(buf, oracle_layers(state, 'decoder'), o, C.rans_decode_with_indexes)
    diff = np.abs(rec.astype(int) - ref_rec.astype(int))
    assert rec.shape == (1024, 1024, 3) and diff.max() <= 1 and (diff > 0).mean() < 1e-3
    codec.close()


@pytest.mark.parametrize('ks,shape', [(3, (3, 1000, 1016)), (5, (2, 520, 488)), (3, (1, 2048, 1040))])
def test_arithmetic_paths_agree_on_large_ragged_tiles(cae, ks, shape):
    """Sizes the CPU oracle cannot reach in seconds: the exact-fp32 kernels and the f16x3 kernels (different tiling,
    different row layouts, persistent first / last layers) must agree with each other on ragged, non-power-of-two
    tiles of the canonical channel counts (k=3 and k=5)."""
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, kernel_size=ks)
    state = synth.synthetic_state(cfg, seed=5)
    models = {}
    for prec in ('fp32', 'f16x3'):
        m = cae.autoencoder_from_state_dict(state)
        for k in ('encoder', 'decoder'):
            m[k].module.precision = prec
        models[prec] = m
    n, h, w = shape
    tiles = torch.from_numpy(synth.uniform_tiles(n, h, w)).cuda()
    y = {p: m['encoder'].module.forward_u8(tiles) for p, m in models.items()}
    assert y['fp32'].shape == (n, 192, -(-h // 16), -(-w // 16))
    np.testing.assert_allclose(y['f16x3'].cpu().numpy(), y['fp32'].cpu().numpy(), rtol=1e-4, atol=1e-4)
    yq = torch.round(y['fp32'])
    rec = {p: m['decoder'].module.forward_u8(yq) for p, m in models.items()}
    diff = (rec['fp32'].int() - rec['f16x3'].int()).abs()
    assert int(diff.max()) <= 1 and float((diff > 0).float().mean()) < 1e-3


@pytest.mark.parametrize('shape', [(5, 5), (6, 9), (16, 16), (17, 33), (33, 17), (64, 5), (8, 130)])
def test_small_and_thin_tiles_against_the_oracle(cae, shape):
    """Smallest legal sizes (every level's input >= 2 for the reflect padding) and thin strips, GDN model, both
    arithmetic paths (parametrised fixture): latents and reconstruction against the CPU oracle."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=8, channels_bn=16, compression_level=3)
    state = synth.synthetic_state(cfg, seed=9)
    model = cae.autoencoder_from_state_dict(state)
    h, w = shape
    tiles = synth.uniform_tiles(2, h, w)
    x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
    y_ref, _ = O.analysis_forward(x, oracle_layers(state, 'encoder'))
    y = model['encoder'].module.forward_u8(torch.from_numpy(tiles).cuda()).cpu()
    assert y.shape == y_ref.shape
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL)
    yq = torch.round(y_ref)
    x_ref, _ = O.synthesis_forward(yq, oracle_layers(state, 'decoder'))
    x_r, _ = model['decoder'](yq.cuda())
    np.testing.assert_allclose(x_r[0].cpu().numpy(), x_ref.numpy(), rtol=RTOL, atol=ATOL * max(1.0, float(x_ref.abs().max())))


def test_error_behaviour(cae):
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=2)
    model = cae.autoencoder_from_state_dict(state)
    with pytest.raises(ValueError):
        model['encoder'](torch.rand(1, 4, 32, 32).cuda())  # wrong channel count
    with pytest.raises(ValueError):
        model['fact_ent'].module.compress(torch.rand(1, 7, 4, 4).cuda())
    with pytest.raises(cae.CaeError):
        model['fact_ent'].module.decompress([b'\x00\x01'], (4, 4))  # truncated stream


def test_fused_quantiser_entry_points_equal_the_unfused_path(cae):
    """cae_analysis_symbols / cae_synthesis_symbols == analysis + quantise / dequantise + synthesis, bit for bit
    (ragged tile size, medians away from zero)."""
    from cnn_autoencoder_amd import synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=4)
    model = cae.autoencoder_from_state_dict(state)
    enc, dec, eb = (model[k].module for k in ('encoder', 'decoder', 'fact_ent'))
    with torch.no_grad():
        eb.quantiles[:, 0, 1] += torch.linspace(-0.4, 0.4, 48, device=eb.quantiles.device)
    eb.update(force=True)
    tiles = torch.from_numpy(synth.uniform_tiles(3, 80, 112)).cuda()
    sym = enc.forward_u8_symbols(tiles, eb)
    assert torch.equal(sym, eb.quantize_symbols(enc.forward_u8(tiles)))
    rec = dec.forward_symbols_u8(sym, eb)
    assert torch.equal(rec, dec.forward_u8(eb.dequantize_symbols(sym)))


def test_one_way_streams_equal_the_simple_calls(cae):
    """SlideCoder.compress_batches / decompress_batches (pipelined encode-only and decode-only streams, host or
    device input) give the payloads / reconstructions of the unpipelined compress() / decompress()."""
    from cnn_autoencoder_amd import slide, synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=6)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    coder = slide.SlideCoder(codec)
    coder.depth = 2
    batches = [synth.uniform_tiles(3, 64, 96) for _ in range(7)]
    batches[3] = torch.from_numpy(batches[3]).cuda()  # device input passes through
    want = [coder.compress(b if isinstance(b, torch.Tensor) else torch.from_numpy(b).cuda()) for b in batches]
    got = list(coder.compress_batches(iter(batches)))
    assert got == want
    rec_want = [coder.decompress(p, 64, 96) for p in want]
    rec_got = list(coder.decompress_batches(iter(want), 64, 96))
    assert len(rec_got) == 7 and all(torch.equal(a, b) for a, b in zip(rec_got, rec_want))
    host = [r.copy() for r in coder.decompress_batches(iter(want), 64, 96, to_host=True)]  # ring buffers: copy at once
    assert len(host) == 7 and all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(host, rec_want))


def test_metrics_match_the_oracle(cae):
    """SSIM / PSNR / RMSE per tile on the GPU against the float64 restatements (ragged size, 3 and 1 channels)."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import metrics, synth
    rng = np.random.default_rng(11)
    for shape in ((3, 70, 101, 3), (2, 64, 64, 1), (1, 7, 9, 3)):
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        x[0] = synth.histo_tile(shape[1], 3, shape[2])[..., :shape[3]]
        y = np.clip(x.astype(int) + rng.integers(-12, 13, shape), 0, 255).astype(np.uint8)
        xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
        ssim = metrics.metric_fun['ssim'](x=xs, x_r=ys).cpu().numpy()
        np.testing.assert_allclose(ssim, [O.ssim_uint8(a, b) for a, b in zip(x, y)], rtol=1e-12)
        psnr = metrics.metric_fun['psnr'](x=xs, x_r=ys).cpu().numpy()
        np.testing.assert_allclose(psnr, [O.psnr_uint8(a, b) for a, b in zip(x, y)], rtol=1e-12)
        rmse = metrics.metric_fun['dist'](x=xs, x_r=ys).cpu().numpy()
        np.testing.assert_allclose(rmse ** 2, [np.mean((a.astype(float) - b) ** 2) for a, b in zip(x, y)], rtol=1e-12)
        if shape[3] == 3:
            de = metrics.metric_fun['delta_cielab'](x=xs, x_r=ys).cpu().numpy()
            np.testing.assert_allclose(de, [O.delta_cielab_uint8(a, b) for a, b in zip(x, y)], rtol=1e-11)
    assert float(metrics.compute_ssim(x=xs, x_r=xs)[0]) == 1.0
    with pytest.raises(ValueError):
        metrics.compute_ssim(x=xs[:, :5], x_r=ys[:, :5])
    with pytest.raises(AssertionError):  # pytorch_msssim's size requirement
        metrics.metric_fun['ms-ssim'](x=xs, x_r=ys)
    for shape in ((2, 200, 171, 3), (1, 256, 320, 1)):
        x = rng.integers(0, 256, shape, dtype=np.uint8)
        x[0] = synth.histo_tile(shape[1], 5, shape[2])[..., :shape[3]]
        y = np.clip(x.astype(int) + rng.integers(-25, 26, shape), 0, 255).astype(np.uint8)
        ms = metrics.metric_fun['ms-ssim'](x=torch.from_numpy(x).cuda(), x_r=torch.from_numpy(y).cuda()).cpu().numpy()
        # tolerance stated: the oracle is float32 torch-CPU, the kernel sums its float32 maps in float64
        np.testing.assert_allclose(ms, [O.ms_ssim_uint8(a, b) for a, b in zip(x, y)], rtol=2e-5)


def test_validation_objective_matches_cpu_restatement(cae):
    """forward_func + GeneralLoss (rate + lambda * 255^2 * MSE, the `valid` half of the training loop) on the HIP
    path against the same formulas on the oracle's torch-CPU tensors."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import criteria, synth
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    state = synth.synthetic_state(cfg, seed=8)
    model = cae.autoencoder_from_state_dict(state)
    x = torch.rand(2, 3, 48, 80)
    with torch.no_grad():
        out = criteria.setup_forward_func()(x.cuda(), model)
        loss = criteria.GeneralLoss(distortion_lambda=0.01)(x.cuda(), out, net=model)
    # oracle: same state, torch-CPU
    y, _ = O.analysis_forward(x, oracle_layers(state, 'encoder'))
    eb = O.EntropyBottleneckOracle(48)
    eb.load({k: v.cpu() for k, v in model['fact_ent'].module.state_dict().items()})
    y_q, p = eb.forward(y)
    x_r, _ = O.synthesis_forward(y_q, oracle_layers(state, 'decoder'))
    rate = -torch.log2(p.double()).sum() / (2 * 48 * 80)
    dist = 255 ** 2 * torch.mean((x_r - x) ** 2)
    assert float(loss['rate_loss']) == pytest.approx(float(rate), rel=1e-4)
    assert float(loss['dist'][0]) == pytest.approx(float(dist), rel=1e-4)
    assert float(loss['loss']) == pytest.approx(float(rate + 0.01 * dist), rel=1e-4)
    assert float(loss['entropy_loss']) == pytest.approx(float(eb.loss()), rel=1e-5)
    assert out['t_pred'] is None and len(out['x_r']) == 3 and out['y_q'].shape == out['p_y'].shape
    # eval-mode modules run the inference kernels also when autograd happens to be enabled (the reference's codec.encode
    # forgets no_grad, _autoencoders.py:539-555): same values, nothing recorded; training is tests/test_train.py
    out2 = criteria.setup_forward_func()(x.cuda(), model)
    assert torch.equal(out2['y'], out['y']) and not out2['y'].requires_grad


def test_tile_sse_exact(cae):
    from cnn_autoencoder_amd import slide, synth
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (3, 48, 80, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (3, 48, 80, 3), dtype=np.uint8)
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=8, channels_bn=16, compression_level=2), seed=1)
    coder = slide.SlideCoder(cae.ConvolutionalAutoencoder(checkpoint=state))
    got = coder.tile_sse(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()).cpu().numpy()
    want = ((a.astype(np.int64) - b.astype(np.int64)) ** 2).reshape(3, -1).sum(axis=1)
    assert np.array_equal(got, want.astype(np.float64))


def test_pipelined_run_equals_simple_roundtrip(cae):
    from cnn_autoencoder_amd import slide, synth
    state = synth.synthetic_state(dict(synth.CANONICAL, channels_net=32, channels_bn=48), seed=2)
    coder = slide.SlideCoder(cae.ConvolutionalAutoencoder(checkpoint=state))
    batches = [torch.from_numpy(synth.uniform_tiles(3, 64, 96, seed=s)).cuda() for s in (1, 2, 3)]
    stats, payloads = coder.run(batches, keep_payloads=True)
    ref = [coder.roundtrip(b) for b in batches]
    assert payloads == [r[0] for r in ref]
    assert torch.equal(stats, torch.cat([r[2] for r in ref]))
    s = slide.slide_summary(slide.gather_stats(stats), 64 * 96)
    assert s['tiles'] == 9 and s['bpp'] > 0 and np.isfinite(s['psnr'])


def _canonical_codec(cae, fit=True, seed=0):
    from cnn_autoencoder_amd import synth
    from oracle import cae_oracle as O
    state = synth.synthetic_state(synth.CANONICAL, seed=seed)
    codec = cae.ConvolutionalAutoencoder(checkpoint=state)
    eb = codec._model['fact_ent'].module
    if fit:
        eb.fit_quantiles()
        eb.update(force=True)
    o = O.EntropyBottleneckOracle(192)
    o.load({k: v.cpu() for k, v in eb.state_dict().items()})
    o.update()
    assert torch.equal(o._quantized_cdf, eb._quantized_cdf.cpu())
    return codec, state, o


def test_cfg2_batch32_256_psnr_bpp_vs_cpu_reference(cae):
    """BASELINE config 2: synthetic 256x256x3 tiles, batch 32, analysis+synthesis on one MI355X; PSNR / bpp
    against the CPU oracle's own compress -> decompress of the same tiles (4 of them, to bound CPU time)."""
    from cnn_autoencoder_amd import slide, synth
    from oracle import c_oracle as C
    from oracle import cae_oracle as O
    codec, state, o = _canonical_codec(cae)
    tiles = np.concatenate([synth.histo_tiles(4, 256), synth.uniform_tiles(28, 256)])
    coder = slide.SlideCoder(codec)
    stats, payloads = coder.run([torch.from_numpy(tiles).cuda()], keep_payloads=True)
    enc_l, dec_l = oracle_layers(state, 'encoder'), oracle_layers(state, 'decoder')
    for i in range(4):
        buf = O.codec_encode(tiles[i], enc_l, o, C.rans_encode_with_indexes)
        rec = O.codec_decode(buf, dec_l, o, C.rans_decode_with_indexes)
        sse_ref = float(((rec.astype(np.float64) - tiles[i]) ** 2).sum())
        bpp_ref, bpp_gpu = 8 * len(buf) / 256 ** 2, 8 * float(stats[i, 0]) / 256 ** 2
        psnr = lambda sse: 10 * np.log10(255.0 ** 2 / (sse / tiles[i].size))
        assert abs(bpp_gpu - bpp_ref) <= 2e-3 * bpp_ref + 1e-3, (bpp_gpu, bpp_ref)
        assert abs(psnr(float(stats[i, 1])) - psnr(sse_ref)) < 1e-3
    assert stats.shape == (32, 3)


def test_cfg3_batch128_256_bitstreams_bit_exact(cae):
    """BASELINE config 3: 128 x 256x256 tiles, full entropy-coded bitstreams on one MI355X, bit-exact against
    the oracle coder on the same latents; the decoder inverts every stream exactly."""
    from cnn_autoencoder_amd import synth
    from oracle import c_oracle as C
    codec, state, o = _canonical_codec(cae, fit=False)
    tiles = np.concatenate([synth.histo_tiles(8, 256, first_index=100), synth.uniform_tiles(120, 256, seed=5)])
    bufs = codec.encode_batch(tiles)
    y = codec._model['encoder'].module.forward_u8(torch.from_numpy(tiles).cuda()).cpu()
    ref = o.compress(y, C.rans_encode_with_indexes)
    assert [b[16:] for b in bufs] == ref
    assert all(struct.unpack('>QQ', b[:16]) == (256, 256) for b in bufs)
    eb = codec._model['fact_ent'].module
    sym = eb.decode_symbols([b[16:] for b in bufs], 16 * 16)
    assert np.array_equal(sym.reshape(128, 192, 16, 16), o.symbols(y).numpy())
    rec = codec.decode_batch(bufs)
    assert rec.shape == (128, 256, 256, 3)


@pytest.mark.parametrize('cfgkw,shape', [(dict(channels_net=96, channels_bn=72, compression_level=2, bias=True), (2, 12, 64)),
                                         (dict(channels_net=40, channels_bn=48, compression_level=3), (1, 5, 9)),
                                         (dict(channels_org=1, channels_net=32, channels_bn=16, compression_level=2), (3, 8, 8)),
                                         (dict(channels_net=128, channels_bn=192, compression_level=4), (2, 4, 6)),
                                         (dict(channels_net=64, channels_bn=48, compression_level=3, act_layer_type=None), (1, 7, 3))])
def test_product_map_form_of_the_last_two_layers(cae, cfgkw, shape, monkeypatch, precision):
    """Codec decode path (uint8 and float without bridges): the second-to-last layer stores the product of its output
    with the last layer's weights and the last layer is a gather (cae_kernels_f16.hpp, pmap; f16x3, k = 3).  Against
    the oracle's truncated reconstruction (<= 1 level, rarely) and against the two-kernel form of the same library;
    channel counts that do not fill their 32-channel tiles (96 -> 4 tiles, 40 -> 2), one image channel, no GDN."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, **cfgkw)
    state = synth.synthetic_state(cfg, seed=17)
    model = cae.autoencoder_from_state_dict(state)
    dec = model['decoder'].module
    n, lh, lw = shape
    torch.manual_seed(2)
    yq = torch.round(torch.randn(n, cfg['channels_bn'], lh, lw) * 3)
    x_ref, _ = O.synthesis_forward(yq, oracle_layers(state, 'decoder'))
    ref8 = (x_ref * 255.0).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
    u8 = dec.forward_u8(yq.cuda()).cpu()
    d = (u8.int() - ref8.int()).abs()
    assert u8.shape == ref8.shape and int(d.max()) <= 1 and float((d > 0).float().mean()) < 5e-3
    xr, _ = dec(yq.cuda(), bridges=False)  # float output through the same form
    np.testing.assert_allclose(xr[0].cpu().numpy(), x_ref.numpy(), rtol=RTOL, atol=ATOL * max(1.0, float(x_ref.abs().max())))
    monkeypatch.setenv('CAE_NO_PMAP', '1')  # the two-kernel form
    u8b = dec.forward_u8(yq.cuda()).cpu()
    monkeypatch.delenv('CAE_NO_PMAP')
    d2 = (u8.int() - u8b.int()).abs()
    assert int(d2.max()) <= 1 and float((d2 > 0).float().mean()) < 5e-3


@pytest.mark.parametrize('ks,shape', [(3, (2, 88, 72)), (3, (1, 50, 130)), (5, (1, 64, 48))])
def test_gdn_wider_than_128_channels_stays_on_f16x3(cae, ks, shape, monkeypatch):
    """channels_net = 192: the fused GDN epilogues hold at most 128 channels in registers, so wider
    layers run the convolution without it and gdn_f16_kernel in place on the split rows -- still the f16x3 arithmetic
    (effective precision 1, no fp32 repeat), same tolerance against the oracle as every other model, ragged sizes
    (rows that do not fill their 32- / 64-pixel groups)."""
    from oracle import cae_oracle as O
    from cnn_autoencoder_amd import synth, _lib
    import ctypes
    monkeypatch.setenv('CAE_PRECISION', 'f16x3')
    cfg = dict(synth.CANONICAL, channels_net=192, channels_bn=48, compression_level=3, kernel_size=ks)
    state = synth.synthetic_state(cfg, seed=23)
    model = cae.autoencoder_from_state_dict(state)
    enc, dec = model['encoder'].module, model['decoder'].module
    assert enc.precision_code() == 1 and dec.precision_code() == 1
    n, h, w = shape
    torch.manual_seed(4)
    x = torch.rand(n, 3, h, w)
    y_ref, _ = O.analysis_forward(x, oracle_layers(state, 'encoder'))
    y = model['encoder'](x.cuda()).cpu()
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=RTOL, atol=ATOL * max(1.0, float(y_ref.abs().max())))
    yq = torch.round(y_ref)
    x_ref, brg_ref = O.synthesis_forward(yq, oracle_layers(state, 'decoder'))
    xr, brg = dec(yq.cuda())
    np.testing.assert_allclose(xr[0].cpu().numpy(), x_ref.numpy(), rtol=RTOL, atol=ATOL * max(1.0, float(x_ref.abs().max())))
    for a, b in zip(brg, brg_ref):
        np.testing.assert_allclose(a.cpu().numpy(), b.numpy(), rtol=RTOL, atol=ATOL * max(1.0, float(b.abs().max())))
    for tr in (enc, dec):
        eff = ctypes.c_int(-1)
        _lib.check(_lib.lib().cae_model_effective_precision(tr._handle.ptr, ctypes.byref(eff)))
        assert eff.value == 1 and tr.fp32_fallbacks == 0


def test_single_layer_fp32_synthesis_reads_only_its_own_planes(built_lib):
    """Regression (randomised sweep, seed 312 case 342): with ONE synthesis layer the fp32 last-layer kernel reads the
    converted latents directly, in 16-channel groups; 72 latent channels were 9 planes of 8 and the kernel read a tenth
    past the buffer -- silent inside the allocator's slack, a GPU memory fault when the buffer ended on a mapping.  Run in
    a child process with the fragment allocator off (every allocation its own mapping), against the CPU replay."""
    import subprocess, sys, os, textwrap
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import cnn_autoencoder_amd as cae
        from test_host import cpu_track
        for bn in (72, 4):
            kw = dict(channels_org=1, channels_net=64, channels_bn=bn, compression_level=1, kernel_size=5, bias=False,
                      act_layer_type='LeakyReLU', batch_norm=False)
            torch.manual_seed(bn)
            dec = cae.Synthesizer(**kw).eval()
            dec.precision = 'fp32'
            yq = torch.round(torch.randn(3, bn, 26, 46) * 3)
            with torch.no_grad():
                ref, _ = cpu_track(dec.synthesis_track, yq, True)
                out, _ = dec.cuda()(yq.cuda())
            torch.cuda.synchronize()
            err = float((out[0].cpu() - ref).abs().max() / max(1.0, float(ref.abs().max())))
            assert err < 1e-4, (bn, err)
        print('ok')
    """) % (ROOT, os.path.join(ROOT, 'tests'))
    env = dict(os.environ, HSA_DISABLE_FRAGMENT_ALLOCATOR='1', CAE_PRECISION='fp32')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'ok' in r.stdout and 'Memory access fault' not in (r.stdout + r.stderr), r.stderr[-800:]
