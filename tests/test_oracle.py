"""CPU tests of the oracle itself: pinned against the reference-generated golden fixtures (conv stacks)
and, for the compressai-side arithmetic that cannot be executed here ("parity unpinned"), against
hand-worked known-answer vectors, invariants and Python-vs-C cross-implementation agreement."""
import json
import os

import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

from conftest import GOLD, golden_state, load_golden, oracle_layers
from oracle import c_oracle as C
from oracle import cae_oracle as O

CASES = ['noact_small_40x56', 'noact_small_37x45', 'gdn_small_40x56', 'gdn_small_37x45', 'gdn_mnist_32x32',
         'gdn_k5bias_48x48', 'gdn_canonical_64x64', 'gdn_canonical_96x80', 'lrelu_bias_small_40x56',
         'relu_small_37x45', 'lrelu_k5_mid_48x48']


@pytest.mark.parametrize('name', CASES)
def test_conv_stacks_match_reference_golden(name):
    """oracle.analysis_forward / synthesis_forward == the reference's Analyzer / Synthesizer outputs."""
    g, cfg = load_golden(name)
    state = golden_state(g, cfg)
    with torch.no_grad():
        x = O.tile_to_input(g['tile'])
        y, outs = O.analysis_forward(x, oracle_layers(state, 'encoder'))
        np.testing.assert_allclose(y.numpy(), g['y'], rtol=0, atol=1e-6)
        for i, t in enumerate(outs):
            td = t.double()
            np.testing.assert_allclose([td.sum().item(), td.abs().sum().item(), (td * td).sum().item()],
                                       g[f'enc_stats_{i}'], rtol=1e-6)
            if f'enc_out_{i}' in g.files:
                np.testing.assert_allclose(t.numpy(), g[f'enc_out_{i}'], rtol=0, atol=1e-6)
        x_r, douts = O.synthesis_forward(torch.round(torch.from_numpy(g['y'])), oracle_layers(state, 'decoder'))
        np.testing.assert_allclose(x_r.numpy(), g['x_r'], rtol=0, atol=1e-6)
        for i, t in enumerate(douts):
            if f'dec_out_{i}' in g.files:
                np.testing.assert_allclose(t.numpy(), g[f'dec_out_{i}'], rtol=0, atol=1e-6)
        u8 = O.output_to_tile(x_r[0])
        assert (np.abs(u8.astype(int) - g['x_r_u8'].astype(int)) > 0).mean() < 1e-4


def test_output_to_tile_truncates():
    x = torch.tensor([[[0.999 / 255 * 255 / 255, 1.5 / 255, -0.2, 2.0]]])  # (1,1,4)
    assert O.output_to_tile(x).reshape(-1).tolist() == [0, 1, 0, 255]


# ---- known-answer vectors (hand-worked) ---------------------------------------------------------
# Notation: L = 2^31, state x starts at L, symbols are popped in reverse order.
#  single_symbol_half : cdf [0,32768,65536]; sym 0: x_max = 2^15*2^32*2^15 = 2^62 > x;
#        x = (2^31/2^15)<<16 + 0 + 0 = 2^32  -> words lo=0x00000000 hi=0x00000001
#  escape_raw0        : sym 1 = max_value -> raw 0, 0 digits: bypass(0): x=(2^31<<4)|0=2^35;
#        main (start 32768, freq 32768): x=(2^35/2^15)<<16 + 0 + 32768 = 2^36+2^15 -> lo 0x00008000 hi 0x10
#  escape_negative    : sym -1 -> raw 1, 1 digit: digit 1: x=2^35+1; count 1: x=(x<<4)|1=2^39+17;
#        main: q=2^24 r=17 -> x=2^40+17+32768 -> lo 0x00008011 hi 0x00000100
#  renormalise_once   : cdf [0,1,65536]; sym 0 (freq 1): x_max=2^47; 1st: x=2^31<<16=2^47;
#        2nd: x>=x_max -> emit lo32(2^47)=0, x=2^15 -> x=2^31; flush lo=0x80000000 hi=0, then the emitted 0
#  two_digit_bypass   : cdf [0,32768,65536], max_value 1; sym 9 >= 1 -> raw = 2(9-1) = 16 = 0x10: two digits, low first
#        (0, 1).  Stack: main, count 2, digit 0, digit 1; popped in reverse: bypass(1): x=(2^31<<4)|1=2^35+1;
#        bypass(0): x=2^39+16; bypass(2): x=2^43+258; main (start 2^15, freq 2^15): q=2^28, r=258 ->
#        x=2^44+258+32768=2^44+0x8102 -> lo 0x00008102 hi 0x00001000
#  seven_digits_with_renormalisation : sym -2^27 -> raw = 2^28-1 (odd: negative), seven digits 0xF.  Seven bypass(15):
#        x = 2^59 + 2^28 - 1; bypass(count 7): x >= x_max = 2^59 -> emit lo32 = 0x0FFFFFFF, x = 2^27 ->
#        x = (2^27<<4)|7 = 2^31+7; main: q = 2^16, r = 7 -> x = 2^32+7+32768 -> lo 0x00008007 hi 1, then the emitted word.
#        (raw is 32-bit upstream and < 2^28 here: at most 8 (7) digits, so the `count >= 15` unary continuation of the
#        digit count can never run; this is the longest count that can be coded.)
#  cdf [1e-9,.75,.25] : rounds to [0,49152,16384], sums [0,0,49152,65536]; entry 0 has freq 0: the smallest freq > 1 is
#        entry 2 (16384), i.e. best_steal > i -> cdf[1..2] += 1 -> [0,1,49153,65536]
#  cdf [1e-9,.5,.5]   : freqs [0,32768,32768]: ties go to the FIRST smallest (strict <) = entry 1 -> cdf[1] += 1
with open(os.path.join(GOLD, 'rans_kat.json')) as f:
    KAT = json.load(f)


@pytest.mark.parametrize('kat', KAT['rans'], ids=[k['name'] for k in KAT['rans']])
@pytest.mark.parametrize('impl', ['python', 'c'])
def test_rans_known_answers(kat, impl):
    enc = O.rans_encode_with_indexes if impl == 'python' else C.rans_encode_with_indexes
    dec = O.rans_decode_with_indexes if impl == 'python' else C.rans_decode_with_indexes
    idx = np.repeat(np.arange(len(kat['cdf'])), kat['hw']).tolist()
    out = enc(kat['symbols'], idx, kat['cdf'], kat['cdf_length'], kat['offset'])
    assert out.hex() == kat['bytes_hex']
    assert list(dec(out, idx, kat['cdf'], kat['cdf_length'], kat['offset'])) == kat['symbols']


@pytest.mark.parametrize('kat', KAT['cdf'])
def test_cdf_known_answers(kat):
    assert O.pmf_to_quantized_cdf(kat['pmf']) == kat['cdf']
    assert C.pmf_to_quantized_cdf(kat['pmf']) == kat['cdf']


pmfs = st.lists(st.floats(min_value=0, max_value=1, allow_nan=False, width=32), min_size=2, max_size=80).filter(
    lambda p: sum(p) > 1e-3)


@settings(max_examples=200, deadline=None)
@given(pmfs)
def test_cdf_invariants_and_agreement(p):
    p = np.asarray(p, dtype=np.float32)
    p = p / p.sum()
    if (np.round(p * 65536) > 1).sum() == 0:
        return
    a = O.pmf_to_quantized_cdf(p.tolist())
    assert a == C.pmf_to_quantized_cdf(p)
    assert a[0] == 0 and a[-1] == 65536 and len(a) == len(p) + 1
    assert all(a[i + 1] > a[i] for i in range(len(a) - 1))


def _random_tables(rng, channels, max_len):
    lens, rows = [], []
    for _ in range(channels):
        n = int(rng.integers(2, max_len + 1))
        p = rng.random(n).astype(np.float32) ** 3 + 1e-4
        rows.append(C.pmf_to_quantized_cdf(p / p.sum()))
        lens.append(n + 1)
    stride = max(lens)
    cdf = np.zeros((channels, stride), dtype=np.int32)
    for i, r in enumerate(rows):
        cdf[i, :len(r)] = r
    off = rng.integers(-8, 3, channels).astype(np.int32)
    return cdf, np.asarray(lens, dtype=np.int32), off


@pytest.mark.parametrize('seed', range(6))
def test_python_and_c_coders_agree_and_round_trip(seed):
    rng = np.random.default_rng(seed)
    channels, hw = int(rng.integers(1, 7)), int(rng.integers(1, 40))
    cdf, lens, off = _random_tables(rng, channels, 24)
    spread = [3, 3, 40, 40, 5000, 2 ** 20][seed]
    sym = rng.integers(-spread, spread + 1, channels * hw).astype(np.int32)
    idx = np.repeat(np.arange(channels), hw).astype(np.int32)
    a = O.rans_encode_with_indexes(sym.tolist(), idx.tolist(), cdf.tolist(), lens.tolist(), off.tolist())
    b = C.rans_encode_with_indexes(sym, idx, cdf, lens, off)
    assert a == b and len(a) % 4 == 0 and len(a) >= 8
    assert O.rans_decode_with_indexes(a, idx.tolist(), cdf.tolist(), lens.tolist(), off.tolist()) == sym.tolist()
    assert C.rans_decode_with_indexes(a, idx, cdf, lens, off) == sym.tolist()


def test_entropy_bottleneck_tables_and_round_trip():
    torch.manual_seed(0)
    eb = O.EntropyBottleneckOracle(12)
    eb.update()
    cdf, lens = eb._quantized_cdf.numpy(), eb._cdf_length.numpy()
    for c in range(12):
        row = cdf[c, :lens[c]]
        assert row[0] == 0 and row[-1] == 65536 and (np.diff(row) > 0).all()
        assert (cdf[c, lens[c]:] == 0).all()
    assert (eb._offset.numpy() == -10).all() and (lens == 23).all()  # init quantiles (-10, 0, 10)
    y = torch.randn(2, 12, 6, 5) * 8
    strings = eb.compress(y, C.rans_encode_with_indexes)
    yq, lik = eb.forward(y)
    assert torch.equal(eb.decompress(strings, (6, 5), C.rans_decode_with_indexes), yq)
    assert (lik >= 1e-9).all() and (lik <= 1).all()
    # likelihood of the quantised value equals the table frequency up to quantisation
    sym = eb.symbols(y)
    c, v = 3, int(sym[0, 3, 0, 0]) - int(eb._offset[3])
    if 0 <= v < lens[3] - 2:
        assert abs((cdf[3, v + 1] - cdf[3, v]) / 65536 - float(lik[0, 3, 0, 0])) < 2e-3


def test_gdn_identity_cases():
    b, g = O.gdn_init_params(5)
    x = torch.randn(1, 5, 3, 3)
    y = O.gdn_forward(x, b, g)
    ref = x / torch.sqrt(1 + 0.1 * x ** 2)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), rtol=1e-5, atol=1e-6)
    z = O.gdn_forward(y, b, g, inverse=True)
    assert z.shape == x.shape and torch.isfinite(z).all()


def test_ssim_restatement_known_answers():
    """skimage is absent: the SSIM restatement is held by properties with closed forms -- identical images give 1,
    a constant offset d between two flat images gives (2 u (u+d) + C1) / (u^2 + (u+d)^2 + C1) exactly (variances 0),
    symmetry, and an independent direct (loop) evaluation of one window."""
    rng = np.random.default_rng(0)
    x = rng.integers(0, 256, (24, 31, 3), dtype=np.uint8)
    y = np.clip(x.astype(int) + rng.integers(-20, 21, x.shape), 0, 255).astype(np.uint8)
    assert O.ssim_uint8(x, x) == pytest.approx(1.0, abs=1e-15)
    assert O.ssim_uint8(x, y) == pytest.approx(O.ssim_uint8(y, x), rel=1e-14)
    flat_a, flat_b = np.full((16, 16, 1), 100, np.uint8), np.full((16, 16, 1), 130, np.uint8)
    c1 = (0.01 * 255) ** 2
    assert O.ssim_uint8(flat_a, flat_b) == pytest.approx((2 * 100 * 130 + c1) / (100 ** 2 + 130 ** 2 + c1), rel=1e-13)
    # one window by hand: 7x7 image -> a single valid... (crop leaves 1x1 at the centre)
    a, b = x[:7, :7, :1].astype(np.float64), y[:7, :7, :1].astype(np.float64)
    ux, uy = a.mean(), b.mean()
    vx, vy = a.var(ddof=1), b.var(ddof=1)
    vxy = ((a - ux) * (b - uy)).sum() / 48.0
    c2 = (0.03 * 255) ** 2
    want = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux ** 2 + uy ** 2 + c1) * (vx + vy + c2))
    assert O.ssim_uint8(x[:7, :7, :1], y[:7, :7, :1]) == pytest.approx(want, rel=1e-12)


def test_delta_cielab_restatement_known_answers():
    """Closed forms: white is L*=100, a*=b*=0 (so black vs white is 100), identical images give 0, and sRGB pure red is
    the textbook (53.24, 80.09, 67.20) for this matrix / white point."""
    white, black = np.full((4, 5, 3), 255, np.uint8), np.zeros((4, 5, 3), np.uint8)
    assert O.delta_cielab_uint8(white, white) == 0.0
    assert O.delta_cielab_uint8(white, black) == pytest.approx(100.0, abs=2e-3)
    red = np.zeros((2, 2, 3), np.uint8)
    red[..., 0] = 255
    # distance red -> black = |Lab(red)| since Lab(black) = 0
    assert O.delta_cielab_uint8(red, np.zeros_like(red)) == pytest.approx(np.sqrt(53.24 ** 2 + 80.09 ** 2 + 67.20 ** 2), abs=0.05)


def test_ms_ssim_restatement_properties():
    """Identical images give exactly 1; the index is symmetric in its arguments and falls as noise grows."""
    rng = np.random.default_rng(3)
    x = rng.integers(0, 256, (176, 200, 3), dtype=np.uint8)
    assert O.ms_ssim_uint8(x, x) == pytest.approx(1.0, abs=1e-6)
    vals = []
    for amp in (4, 16, 64):
        y = np.clip(x.astype(int) + rng.integers(-amp, amp + 1, x.shape), 0, 255).astype(np.uint8)
        vals.append(O.ms_ssim_uint8(x, y))
        assert O.ms_ssim_uint8(y, x) == pytest.approx(vals[-1], rel=1e-5)
    assert 1 > vals[0] > vals[1] > vals[2] > 0
