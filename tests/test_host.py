"""CPU tests of the product's host side: the C ABI surface, the host entropy coder inside
libcae_hip.so (checked against the oracle), the nn.Module / codec boundary and the slide driver.
No device entry point is called here (there is no GPU in the build container)."""
import ctypes
import json
import os
import re
import struct

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from conftest import GOLD, ROOT, load_golden
from oracle import c_oracle as C
from oracle import cae_oracle as O


@pytest.fixture(scope='module')
def cae(built_lib):
    import cnn_autoencoder_amd as cae
    return cae


def test_abi_exports_every_declared_symbol(cae):
    from cnn_autoencoder_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'cae_hip.h')).read()
    declared = set(re.findall(r'\b(cae_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    L = ctypes.CDLL(cae.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert _lib.lib().cae_version() == 1


def test_no_product_module_touches_the_oracle():
    pkg = os.path.join(ROOT, 'cnn_autoencoder_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.cpp', '.hpp', '.h')) or f == 'Makefile':
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.lower().replace('# oracle-free', ''), f'{f} mentions the oracle'


def test_missing_gpu_fails_loudly(cae):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    enc = cae.Analyzer(3, 8, 16, 3, act_layer_type='GDN')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        enc(torch.rand(1, 3, 16, 16))
    eb = cae.EntropyBottleneck(4)
    eb.update()
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        eb.compress(torch.rand(1, 4, 2, 2))


# ---- quantised CDF ------------------------------------------------------------------------------
pmfs = st.lists(st.floats(min_value=0, max_value=1, allow_nan=False, width=32), min_size=2, max_size=60).filter(
    lambda p: sum(p) > 1e-3)


@settings(max_examples=150, deadline=None)
@given(pmfs)
def test_product_cdf_equals_oracle(p):
    from cnn_autoencoder_amd.entropy import pmf_to_quantized_cdf
    p = np.asarray(p, dtype=np.float32)
    p = p / p.sum()
    if (np.round(p * 65536) > 1).sum() == 0:
        return
    assert pmf_to_quantized_cdf(torch.from_numpy(p)).tolist() == O.pmf_to_quantized_cdf(p.tolist())


def test_product_cdf_rejects_bad_pmf(cae):
    from cnn_autoencoder_amd.entropy import pmf_to_quantized_cdf
    with pytest.raises(ValueError):
        pmf_to_quantized_cdf(torch.tensor([0.5, float('nan')]))
    with pytest.raises(ValueError):
        pmf_to_quantized_cdf(torch.tensor([0.5, -0.1]))
    with pytest.raises(ValueError):
        pmf_to_quantized_cdf(torch.tensor([0.0, 0.0]))


# ---- host rANS coder ----------------------------------------------------------------------------
class _Tables:
    """Drives cae_rans_{encode,decode}_batch with explicit integer tables."""

    def __init__(self, cdf, lens, off):
        from cnn_autoencoder_amd import _lib
        self._lib = _lib
        self.cdf = np.ascontiguousarray(cdf, dtype=np.int32)
        self.lens = np.ascontiguousarray(lens, dtype=np.int32)
        self.off = np.ascontiguousarray(off, dtype=np.int32)
        self.C = self.cdf.shape[0]
        self.h = _lib.Handle(1, 1, self.C, 1, 3)
        med = np.zeros(self.C, dtype=np.float32)
        _lib.check(_lib.lib().cae_model_set_entropy(self.h.ptr, self.C, self.cdf.shape[1], self.cdf.ctypes.data,
                                                    self.lens.ctypes.data, self.off.ctypes.data, med.ctypes.data))

    def encode(self, sym, threads=0):
        sym = np.ascontiguousarray(sym, dtype=np.int32)
        n, hw = sym.shape[0], sym.shape[2]
        bufs = (ctypes.c_void_p * n)()
        lens = (ctypes.c_size_t * n)()
        self._lib.check(self._lib.lib().cae_rans_encode_batch(self.h.ptr, sym.ctypes.data, n, hw, bufs, lens, threads))
        out = [ctypes.string_at(bufs[i], lens[i]) for i in range(n)]
        for i in range(n):
            self._lib.lib().cae_free(bufs[i])
        return out

    def decode(self, strings, hw, threads=0):
        n = len(strings)
        bufs = (ctypes.c_char_p * n)(*strings)
        lens = (ctypes.c_size_t * n)(*[len(s) for s in strings])
        sym = np.empty((n, self.C, hw), dtype=np.int32)
        self._lib.check(self._lib.lib().cae_rans_decode_batch(self.h.ptr, bufs, lens, n, hw, sym.ctypes.data, threads))
        return sym

    def oracle_encode(self, sym_one):
        idx = np.repeat(np.arange(self.C), sym_one.shape[1]).astype(np.int32)
        return C.rans_encode_with_indexes(sym_one.reshape(-1), idx, self.cdf, self.lens, self.off)


def test_product_coder_known_answers(cae):
    import json
    kat = json.load(open(os.path.join(GOLD, 'rans_kat.json')))['rans']
    for k in kat:
        t = _Tables(k['cdf'], k['cdf_length'], k['offset'])
        sym = np.asarray(k['symbols'], dtype=np.int32).reshape(1, len(k['cdf']), k['hw'])
        assert t.encode(sym)[0].hex() == k['bytes_hex'], k['name']
        assert np.array_equal(t.decode([bytes.fromhex(k['bytes_hex'])], k['hw']), sym)


@pytest.mark.parametrize('seed,spread', [(0, 2), (1, 6), (2, 40), (3, 5000), (4, 2 ** 20), (5, 2 ** 26)])
def test_product_coder_bit_exact_vs_oracle(cae, seed, spread):
    from test_oracle import _random_tables
    rng = np.random.default_rng(100 + seed)
    channels, hw, n = int(rng.integers(1, 9)), int(rng.integers(1, 64)), 5
    cdf, lens, off = _random_tables(rng, channels, 40)
    t = _Tables(cdf, lens, off)
    sym = rng.integers(-spread, spread + 1, (n, channels, hw)).astype(np.int32)
    for threads in (1, 3):
        out = t.encode(sym, threads)
        assert out == [t.oracle_encode(sym[i]) for i in range(n)]
        assert np.array_equal(t.decode(out, hw, threads), sym)


def test_out_of_range_symbols_are_rejected_not_hung(cae):
    t = _Tables([[0, 32768, 65536]], [3], [0])
    with pytest.raises(ValueError, match='codable range'):
        t.encode(np.asarray([[[2 ** 29]]], dtype=np.int32))
    with pytest.raises(ValueError):
        C.rans_encode_with_indexes(np.asarray([2 ** 29], dtype=np.int32), np.zeros(1, dtype=np.int32),
                                   np.asarray([[0, 32768, 65536]], dtype=np.int32), [3], [0])


def test_reciprocal_division_is_exact_for_every_frequency(cae):
    """The encoder replaces x / freq by a fixed-point reciprocal multiply: compare with the oracle's
    plain division for every frequency 1..65535 (two-bin rows [0, f, 65536])."""
    freqs = np.arange(1, 65536, dtype=np.int64)
    rng = np.random.default_rng(7)
    for block in np.array_split(freqs, 64):
        cdf = np.zeros((len(block), 3), dtype=np.int32)
        cdf[:, 1] = block
        cdf[:, 2] = 65536
        lens = np.full(len(block), 3, dtype=np.int32)
        off = np.zeros(len(block), dtype=np.int32)
        t = _Tables(cdf, lens, off)
        sym = rng.integers(0, 2, (1, len(block), 24)).astype(np.int32)  # value 1 = escape bin (freq 65536-f)
        assert t.encode(sym, 1)[0] == t.oracle_encode(sym[0])


def test_corrupt_and_short_streams_raise(cae):
    rng = np.random.default_rng(0)
    from test_oracle import _random_tables
    cdf, lens, off = _random_tables(rng, 3, 10)
    t = _Tables(cdf, lens, off)
    sym = rng.integers(-3, 4, (1, 3, 50)).astype(np.int32)
    s = t.encode(sym)[0]
    with pytest.raises(cae.CaeError, match='bitstream'):
        t.decode([s[:4]], 50)
    with pytest.raises(cae.CaeError, match='bitstream'):
        t.decode([b'\x00' * 8], 50)  # state 0 keeps pulling words past the end
    with pytest.raises(ValueError):
        t.encode(sym[:0])


# ---- nn.Module / checkpoint / codec boundary ----------------------------------------------------
def test_hot_kernels_keep_their_accumulators_in_registers(cae):
    """No scratch (private) memory in the MFMA kernels of the canonical model: an epilogue change that makes the
    compiler index the accumulator array dynamically moves 64-256 registers per lane into scratch without
    reporting a single spill (measured twice in round 1: 3x slower layer).  tools/scratch_check.py."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('scratch_check', os.path.join(ROOT, 'tools', 'scratch_check.py'))
    sc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sc)
    if not os.path.exists(sc.READELF):
        pytest.skip('llvm-readelf not available')
    table = sc.kernel_table(cae.LIB_PATH)
    names = list(table)
    nice = dict(zip(sc.demangle(names), names))
    hot = ['conv_s2_f16_kernel<3, 4, true, 2,', 'conv_s2_f16_kernel<3, 6, false, 2,', 'conv_s2_f16_kernel<5, 4, true, 2,',
           'conv_s2_f16_kernel<3, 4, false, 1,',
           'deconv_s2_f16_kernel<3, 4, 8, 1, true>', 'deconv_s2_f16_kernel<3, 4, 8, 1, false>',
           'deconv_last_f16_kernel<3, 8, 3>', 'conv_s2_kernel<3, 4, 4, true, 2, false, false>',
           'conv_s2_kernel<3, 6, 4, false, 2, false, false>', 'deconv_last_kernel<3', 'likelihood_kernel<3>']
    for h in hot:
        found = [k for k in nice if h in k]
        assert found, f'{h} not in the library'
        for k in found:
            priv, spill = table[nice[k]]
            assert priv == 0 and spill == 0, f'{k}: {priv} B scratch, {spill} spilled VGPRs'


def test_init_and_state_dict_keys_match_reference_fixture(cae):
    g = np.load(os.path.join(GOLD, 'ref_init_seed0.npz'))
    torch.manual_seed(0)
    enc = cae.Analyzer(3, 8, 16, 3, act_layer_type='GDN')
    dec = cae.Synthesizer(3, 8, 16, 3, act_layer_type='GDN')
    sd = {**{'encoder/' + k: v for k, v in enc.state_dict().items()},
          **{'decoder/' + k: v for k, v in dec.state_dict().items()}}
    for k in g.files:  # every tensor the reference's modules hold, bit for bit under the same seed
        assert np.array_equal(g[k], sd[k].numpy()), k
    extra = [k for k in sd if k not in g.files]
    assert all(re.search(r'(beta|gamma)_reparam\.(pedestal|lower_bound\.bound)$', k) for k in extra)
    assert len(enc.analysis_track) == 3 and len(dec.synthesis_track) == 3 and dec.rec_level == 3


def test_unsupported_variants_say_so(cae):
    with pytest.raises(NotImplementedError):
        cae.Analyzer(3, 8, 16, 3, kernel_size=7)
    # residual units ride the split-f16 kernels (stages wider than 128 channels detour through fp32 inside the library)
    assert cae.Analyzer(3, 8, 16, 3, use_residual=True, act_layer_type='GDN').precision_code() == 1
    assert cae.Analyzer(3, 160, 16, 3, use_residual=True, act_layer_type='GDN').precision_code() == 1
    with pytest.raises(ValueError, match='not supported'):  # the reference's own message (_autoencoders.py:32)
        cae.Analyzer(3, 8, 16, 3, act_layer_type='LeakyRelU')
    with pytest.raises(ValueError, match='divisible by groups'):  # nn.Conv2d's own condition (3 -> 8, groups=3)
        cae.Analyzer(3, 8, 16, 3, groups=True)
    # LeakyReLU / ReLU units: reference module order -> state-dict indices model.0 (s1 conv) / model.2 (s2 conv)
    a = cae.Analyzer(3, 8, 16, 3, act_layer_type='LeakyReLU', bias=True)
    assert list(a.state_dict()) == [f'analysis_track.{i}.model.{j}.{p}' for i, js in ((0, (0, 2)), (1, (0, 2)), (2, (0,)))
                                    for j in js for p in ('weight', 'bias')]
    assert a.precision_code() == 1  # (round 3: on the split-f16 kernels like the GDN model)
    assert cae.Synthesizer(3, 160, 16, 3, act_layer_type='ReLU').precision_code() == 0  # > 128 channels with an activation
    assert cae.Synthesizer(3, 8, 16, 3, multiscale_analysis=True).precision_code() == 1  # colour layers: <= 32 image channels
    assert cae.Synthesizer(40, 8, 16, 3, multiscale_analysis=True).precision_code() == 0
    with pytest.raises(NotImplementedError, match='training mode'):  # BatchNorm folds in eval mode only
        cae.Analyzer(3, 8, 16, 3, batch_norm=True).train().analysis_track[0].effective_main()


VARIANTS = ['var_bn_gdn_40x56', 'var_bn_lrelu_bias_37x45', 'var_expansion2_gdn_48x48', 'var_groups_relu_40x40',
            'var_groups_k5_32x48', 'var_multiscale_lrelu_bias_40x56', 'var_multiscale_gdn_k5_48x48',
            'var_res_gdn_40x56', 'var_res_lrelu_bn_bias_37x45', 'var_res_none_k5_48x48', 'var_res_relu_mid_32x32']


def variant_modules(cae, g, cfg):
    """Our Analyzer / Synthesizer for a variant fixture, loaded from the reference's state dict."""
    kw = {k: v for k, v in cfg.items() if k != 'seed'}
    enc, dec = cae.Analyzer(**kw), cae.Synthesizer(**kw)
    for mod, part in ((enc, 'encoder/'), (dec, 'decoder/')):
        sd = {k[len(part):]: torch.from_numpy(np.asarray(g[k])) for k in g.files if k.startswith(part)}
        res = mod.load_state_dict(sd, strict=False)  # same keys, same shapes as the reference module ...
        assert res.unexpected_keys == []
        # ... except the constant buffers of compressai's GDN parametrisation, which the fixture generator's stand-in
        # GDN does not carry (oracle/gen_golden.py)
        assert all('_reparam.' in k for k in res.missing_keys), res.missing_keys
    return enc.eval(), dec.eval()


def cpu_track(units, x, synthesis):
    """The device's view of a track replayed with torch-CPU ops: per unit the folded stride-1 stages
    (y = post_act(act_or_gdn(conv(x)) [+ unit input])) and the folded strided layer, exactly the tensors
    _Track._sync uploads.  -> (output, per-unit outputs)."""
    F = torch.nn.functional

    def act(v, code):
        return v if code == 0 else (F.leaky_relu(v, 0.01) if code == 1 else F.relu(v))

    def conv_s1(v, w, b):
        k = w.shape[-1]
        if synthesis:
            return F.conv_transpose2d(v, w, b, stride=1, padding=k // 2)
        return F.conv2d(F.pad(v, (k // 2,) * 4, mode='reflect'), w, b)

    outs = []
    for u in units:
        unit_in = x
        if hasattr(u, 'stages'):
            stages = u.stages()
        elif u.pre is not None:
            w, b = u.effective_pre()
            stages = [dict(weight=w, bias=b, beta=None, gamma=None, act=u.act_code, add_residual=0, post_act=0)]
        else:
            stages = []
        for sg in stages:
            t = conv_s1(x, sg['weight'], sg['bias'])
            if sg['beta'] is not None:  # effective (re-parametrised) values
                norm = F.conv2d(t * t, sg['gamma'][:, :, None, None], sg['beta'])
                t = t * (torch.sqrt(norm) if synthesis else torch.rsqrt(norm))
            else:
                t = act(t, sg['act'])
            if sg['add_residual']:
                t = t + unit_in
            x = act(t, sg['post_act'])
        w, b = u.effective_main()
        x = O.deconv_s2(x, w, b) if synthesis else O.reflect_conv_s2(x, w, b)
        if u.gdn is not None:
            x = O.gdn_forward(x, u.gdn.beta.detach(), u.gdn.gamma.detach(), synthesis, 1e-6)
        else:
            x = act(x, u.act_code)
        outs.append(x)
    return x, outs


@pytest.mark.parametrize('name', VARIANTS)
def test_variant_state_dicts_and_folding_match_the_reference(cae, name):
    """BatchNorm / groups / channel-expansion variants: the reference's state dict loads strictly, and the folded
    dense layers (what is uploaded to the GPU) reproduce the reference's eval-mode outputs in the CPU oracle."""
    g, cfg = load_golden(name)
    enc, dec = variant_modules(cae, g, cfg)
    x = O.tile_to_input(g['tile'])
    y, _ = cpu_track(enc.analysis_track, x, False)
    np.testing.assert_allclose(y.numpy(), g['y'], rtol=2e-5, atol=2e-5)
    x_r, brg = cpu_track(dec.synthesis_track, torch.round(torch.from_numpy(g['y'])), True)
    np.testing.assert_allclose(x_r.numpy(), g['x_r_0'], rtol=2e-5, atol=2e-5)
    assert [tuple(t.shape) for t in brg] == [tuple(g[f'brg_shape_{i}']) for i in range(len(brg))]
    if cfg.get('multiscale_analysis'):  # colour layers: stride-1 reflect convolutions on the intermediate features
        L, k = cfg['compression_level'], cfg['kernel_size']
        for i, layer in enumerate(list(dec.color_layers)[:-1]):
            conv = layer[0]
            pad = torch.nn.functional.pad(brg[i], (k // 2,) * 4, mode='reflect')
            ref = torch.nn.functional.conv2d(pad, conv.dense_weight(), None if conv.bias is None else conv.bias.detach())
            np.testing.assert_allclose(ref.numpy(), g[f'x_r_{L - 1 - i}'], rtol=2e-5, atol=2e-5)


def test_checkpoint_schema_round_trip(cae, tmp_path):
    from cnn_autoencoder_amd import synth
    cfg = dict(synth.CANONICAL, channels_net=8, channels_bn=16, compression_level=3)
    state = synth.synthetic_state(cfg, seed=3)
    model = cae.autoencoder_from_state_dict(state)
    assert set(model) == {'encoder', 'decoder', 'fact_ent'}
    for k in model:
        assert isinstance(model[k], torch.nn.DataParallel) and not model[k].training
    fe = model['fact_ent'].module
    assert fe.channels == 16 and fe.filters == (3, 3, 3, 3) and fe._quantized_cdf.shape[0] == 16
    # what utils.checkpoint writes: args + module state_dicts (incl. the derived integer buffers)
    ckpt = dict(cfg, encoder=model['encoder'].module.state_dict(), decoder=model['decoder'].module.state_dict(),
                fact_ent=fe.state_dict())
    path = str(tmp_path / 'last_ckpt.pth')
    torch.save(ckpt, path)
    again = cae.autoencoder_from_state_dict(path)
    assert torch.equal(again['fact_ent'].module._quantized_cdf, fe._quantized_cdf)
    for a, b in zip(again['encoder'].module.parameters(), model['encoder'].module.parameters()):
        assert torch.equal(a, b)
    # only some modules enabled
    only = cae.autoencoder_from_state_dict(dict(state, enabled_modules=['decoder']))
    assert set(only) == {'decoder'}


def test_entropy_module_matches_oracle_on_cpu(cae):
    torch.manual_seed(1)
    eb = cae.EntropyBottleneck(10, filters=[3] * 4).eval()
    eb.update(force=True)
    o = O.EntropyBottleneckOracle(10)
    o.load(eb.state_dict())
    o.update()
    assert torch.equal(o._quantized_cdf, eb._quantized_cdf)
    assert torch.equal(o._cdf_length, eb._cdf_length) and torch.equal(o._offset, eb._offset)
    x = torch.randn(2, 10, 4, 3) * 7
    a, b = eb(x)
    c, d = o.forward(x)
    assert torch.equal(a, c) and torch.equal(b, d)
    assert torch.allclose(eb.loss(), o.loss())
    eb.train()
    yn, _ = eb(x)
    assert ((yn - x).abs() <= 0.5).all() and not torch.equal(yn, x)
    eb.fit_quantiles()
    assert float(eb.loss()) < 1e-2
    eb.update(force=True)
    assert int(eb._cdf_length.min()) > 100  # support widened to the 1e-9 tails
    # host coder of the module (no GPU needed): bit-exact vs the oracle coder
    o.load(eb.state_dict())
    o.update()
    sym = torch.round(x * 9).int().reshape(2, 10, -1).numpy()
    strings = eb.encode_symbols(sym)
    idx = np.repeat(np.arange(10), sym.shape[2])
    for i in range(2):
        assert strings[i] == C.rans_encode_with_indexes(sym[i].reshape(-1), idx, o._quantized_cdf.numpy(),
                                                        o._cdf_length.numpy(), o._offset.numpy())
    assert np.array_equal(eb.decode_symbols(strings, sym.shape[2]), sym)


def test_density_upload_validates_filters(cae):
    """cae_model_set_density (host side only): widths up to 8 are packed, wider ones refuse."""
    eb = cae.EntropyBottleneck(6, filters=(2, 8, 3)).eval()
    eb._sync_density()  # builds CDFs, packs the effective parameters; no device work
    wide = cae.EntropyBottleneck(6, filters=(9,)).eval()
    with pytest.raises(cae.CaeError, match='wider than 8'):
        wide._sync_density()
    if not torch.cuda.is_available():
        with torch.no_grad(), pytest.raises(RuntimeError, match='no CPU fallback'):
            eb.rate_bits(torch.zeros(1, 6, 2, 2))


def test_entropy_state_dict_loads_integer_buffers(cae):
    torch.manual_seed(2)
    a = cae.EntropyBottleneck(6)
    a.update()
    b = cae.EntropyBottleneck(6)
    b.load_state_dict(a.state_dict())  # strict: sizes of the integer buffers differ before loading
    assert torch.equal(b._quantized_cdf, a._quantized_cdf) and torch.equal(b._cdf_length, a._cdf_length)
    assert 'likelihood_lower_bound.bound' in a.state_dict() and 'target' in a.state_dict()


def test_bottleneck_codec_config_round_trip(cae):
    torch.manual_seed(4)
    eb = cae.EntropyBottleneck(8, filters=[3] * 4)
    codec = cae.ConvolutionalAutoencoderBottleneck(channels_bn=8, fact_ent=eb)
    cfg = codec.get_config()
    assert cfg['id'] == 'cae_bn' and set(cfg) == {'id', 'filters', 'channels_bn', 'fact_ent_checkpoint'}
    import json
    clone = cae.ConvolutionalAutoencoderBottleneck.from_config(json.loads(json.dumps(cfg)))
    assert torch.equal(clone._fact_ent._quantized_cdf.cpu(), codec._fact_ent._quantized_cdf.cpu())
    assert cae.ConvolutionalAutoencoder.codec_id == 'cae'


# ---- slide driver -------------------------------------------------------------------------------
@settings(max_examples=100, deadline=None)
@given(st.integers(1, 16), st.integers(0, 5000))
def test_tile_range_partitions_the_slide(world, n_tiles):
    from cnn_autoencoder_amd import slide
    ranges = [slide.tile_range(r, world, n_tiles) for r in range(world)]
    assert ranges[0][0] == 0 and ranges[-1][1] == n_tiles
    assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in ranges]
    assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_slide_summary_math():
    from cnn_autoencoder_amd import slide
    with pytest.raises(ValueError):
        slide.tile_range(3, 2, 10)
    st_ = slide.tile_stats([1000, 3000], [100.0, 300.0], 64 * 64 * 3)
    s = slide.slide_summary(slide.gather_stats(st_), 64 * 64)
    assert s['tiles'] == 2 and s['bytes'] == 4000
    assert abs(s['bpp'] - 8 * 4000 / (2 * 64 * 64)) < 1e-12
    mse = 400.0 / (2 * 64 * 64 * 3)
    assert abs(s['psnr'] - 10 * np.log10(255 ** 2 / mse)) < 1e-9


def test_untrusted_metadata_is_not_unpickled(cae, tmp_path):
    """fact_ent_checkpoint blobs and checkpoint paths arrive through .zarray metadata (Codec.from_config): only
    tensors / plain containers are loaded, anything else is refused (weights_only)."""
    import base64
    import io
    import pickle

    class Evil:
        def __reduce__(self):
            return (os.system, ('true',))

    blob = base64.b64encode(pickle.dumps(Evil())).decode('ascii')
    with pytest.raises((ValueError, pickle.UnpicklingError)):
        cae.ConvolutionalAutoencoderBottleneck(channels_bn=4, filters=[3, 3], fact_ent_checkpoint={'quantiles': blob})
    buf = io.BytesIO()
    torch.save({'not': 'a tensor'}, buf)
    with pytest.raises(ValueError, match='expected a tensor'):
        cae.ConvolutionalAutoencoderBottleneck._bytes2tensor(base64.b64encode(buf.getvalue()).decode('ascii'))
    path = str(tmp_path / 'evil.pth')
    with open(path, 'wb') as f:
        pickle.dump({'channels_bn': 4, 'x': Evil()}, f)
    with pytest.raises((ValueError, pickle.UnpicklingError)):
        cae.codec._load_checkpoint(path)


def test_likelihood_forms_agree_except_where_float_rounding_decides(cae):
    """The two floating-point forms of the bin probability (entropy.LIKELIHOOD_FORMS: 'plain' = compressai >= 1.2.x as
    recalled, 'sign_trick' = older) are equal in exact arithmetic.  On the canonical model they give the same CDF row
    for almost every channel; where one pmf entry rounds to a different 16-bit frequency the stealing loop of
    pmf_to_quantized_cdf reshuffles that row, so bitstreams of the two forms are NOT interchangeable: the form is a
    named option, and product and oracle agree on each."""
    from cnn_autoencoder_amd import synth
    from oracle import cae_oracle as O
    state = synth.synthetic_state(synth.CANONICAL, seed=0)
    tables = {}
    for form in cae.entropy.LIKELIHOOD_FORMS:
        eb = cae.EntropyBottleneck(192, filters=[3] * 4, likelihood_form=form)
        eb.load_state_dict(state['fact_ent'], strict=False)
        eb.fit_quantiles()
        eb.update(force=True)
        o = O.EntropyBottleneckOracle(192, likelihood_form=form)
        o.load(eb.state_dict())
        o.update()
        assert torch.equal(o._quantized_cdf, eb._quantized_cdf) and torch.equal(o._cdf_length, eb._cdf_length)
        cdf = eb._quantized_cdf
        for c in range(192):
            row = cdf[c, :int(eb._cdf_length[c])]
            assert row[0] == 0 and row[-1] == 65536 and bool((row[1:] > row[:-1]).all())
        tables[form] = cdf
        # same values where float noise cannot matter: in-support probabilities of the first channel
        v = torch.linspace(-3, 3, 7).view(1, 1, -1).repeat(192, 1, 1)
        tables[form + '_p'] = cae.entropy._likelihood(eb._params(cpu=True), 4, v, form)[0]
    np.testing.assert_allclose(tables['plain_p'].numpy(), tables['sign_trick_p'].numpy(), rtol=2e-6, atol=3e-7)  # (plain: cancellation near sigmoid = 1)
    differing_rows = int((tables['plain'] != tables['sign_trick']).any(dim=1).sum())
    assert 0 < differing_rows <= 10, f'{differing_rows} of 192 CDF rows differ between the two forms'
    with pytest.raises(ValueError):
        cae.EntropyBottleneck(4, likelihood_form='other')


def test_training_objective_matches_the_reference_loss_values():
    """criteria.GeneralLoss / RateLoss / DistMSELoss / setup_loss against values produced by the REFERENCE's own classes
    (src/models/criteria/_lossutils.py:5-109, _ratedist.py:45-63; fixture written by `oracle/gen_golden.py loss`), and
    the oracle's restatement of the same objective (oracle/train_oracle.rd_loss) against both."""
    import torch
    from cnn_autoencoder_amd import criteria
    from oracle import train_oracle as T
    g = np.load(os.path.join(GOLD, 'ref_loss_ratemse.npz'))
    x, x_r, p_y = (torch.from_numpy(g[k]) for k in ('x', 'x_r', 'p_y'))
    aux = torch.tensor(float(g['aux']))

    class _Fe:
        def loss(self):
            return aux

    class _Wrapped:
        module = _Fe()

    assert float(criteria.RateLoss()(x=x, p_y=p_y)['rate_loss']) == pytest.approx(float(g['rate_only']), rel=1e-6)
    assert float(criteria.DistMSELoss()(x=x, x_r=[x_r])['dist'][0]) == pytest.approx(float(g['mse_only']), rel=1e-6)
    for lam, tag in ((0.01, '0p01'), (0.1, '0p1')):
        for crit in (criteria.GeneralLoss(distortion_lambda=lam), criteria.setup_loss('RateMSE', distortion_lambda=lam)):
            ld = crit(inputs=x, outputs=dict(x_r=[x_r, None], p_y=p_y, y=None), net={'fact_ent': _Wrapped()})
            assert sorted(ld.keys()) == json.loads(bytes(g['keys_json']).decode())
            for key in ('loss', 'dist_loss', 'rate_loss', 'entropy_loss'):
                assert float(ld[key]) == pytest.approx(float(g[f'{key}_{tag}']), rel=1e-6), key
            assert float(ld['dist'][0]) == pytest.approx(float(g[f'dist0_{tag}']), rel=1e-6)
        loss_o, dist_o, rate_o = T.rd_loss(x, x_r, p_y, lam)
        assert float(loss_o) == pytest.approx(float(g[f'loss_{tag}']), rel=1e-6)
    with pytest.raises(NotImplementedError):
        criteria.setup_loss('RateMSSSIM')


@pytest.mark.parametrize('name', ['var_res_gdn_40x56', 'var_res_none_k5_48x48', 'var_res_relu_mid_32x32'])
def test_residual_training_restatement_is_pinned_by_the_reference_fixtures(cae, name):
    """oracle/train_oracle.residual_track (the checker of the residual units' gradients, tests/test_train.py) in plain fp32
    reproduces the outputs the REFERENCE's own ResidualDownsamplingUnit / ResidualUpsamplingUnit stacks produced for the
    variant fixtures (unit wiring, activation placement, stride-1 transposed convolutions)."""
    from conftest import load_golden, residual_oracle_units
    from oracle import train_oracle as T
    g, cfg = load_golden(name)
    enc, dec = variant_modules(cae, g, cfg)
    act = cfg.get('act_layer_type')
    act_name = act if act in ('LeakyReLU', 'ReLU') else None
    x = torch.from_numpy(g['tile']).permute(2, 0, 1).unsqueeze(0).float() / 255.0
    with torch.no_grad():
        units, _ = residual_oracle_units(enc.analysis_track, act_name)
        y = T.residual_track(x, units, False, bf16=False)
        np.testing.assert_allclose(y.numpy(), g['y'], rtol=1e-4, atol=1e-4)
        units, _ = residual_oracle_units(dec.synthesis_track, act_name)
        x_r = T.residual_track(torch.round(torch.from_numpy(g['y'])), units, True, bf16=False)
        scale = max(1.0, float(np.abs(g['x_r_0']).max()))
        np.testing.assert_allclose(x_r.numpy(), g['x_r_0'], rtol=1e-4, atol=1e-4 * scale)
