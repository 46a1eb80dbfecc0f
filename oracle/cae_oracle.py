"""CPU oracle for the convolutional-autoencoder compress/decompress hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``cnn_autoencoder_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

What is restated here, and what pins it:

* Analysis / synthesis conv stacks (reference ``src/models/tasks/_autoencoders.py``
  :53-101 DownsamplingUnit, :177-227 UpsamplingUnit, :307-361 Analyzer, :364-455
  Synthesizer) -- restated with ``torch.nn.functional`` on CPU and PINNED against
  golden tensors emitted by the reference's own classes (``oracle/gen_golden.py``,
  fixtures under ``tests/golden/``).
* GDN / IGDN, the factorized EntropyBottleneck (likelihood, CDF construction,
  quantiser), ``pmf_to_quantized_cdf`` and the rANS coder live in the third-party
  package ``compressai`` (``requirements.txt:25``: ``compressai>=1.2.4``, un-pinned,
  not vendored, absent from this image).  They are restated from the published
  algorithm and anchored on the reference call sites
  ``_autoencoders.py:29-30`` (GDN), ``:476-477`` (EntropyBottleneck ctor), ``:502``
  (update), ``:549-551`` (compress), ``:568-571`` (decompress).
  **PARITY UNPINNED** for these: the reference holds no tests or vectors for them
  and ``compressai`` cannot be executed here.  They are cross-checked by invariants,
  hand-worked known-answer vectors (``tests/golden/rans_kat.json``) and by agreement
  between three independent implementations (this file, ``oracle/rans64_oracle.c``
  and the HIP/C++ product).
"""
from __future__ import annotations

import math
import struct
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

# --------------------------------------------------------------------------------------
# GDN (compressai.layers.GDN; reference call site _autoencoders.py:29-30)
# --------------------------------------------------------------------------------------

REPARAM_OFFSET = 2.0 ** -18
PEDESTAL = REPARAM_OFFSET ** 2  # 2**-36


def nonneg_init(x: torch.Tensor) -> torch.Tensor:
    """NonNegativeParametrizer.init: sqrt(max(x + pedestal, pedestal))."""
    ped = torch.tensor([PEDESTAL], dtype=x.dtype)
    return torch.sqrt(torch.max(x + ped, ped))


def nonneg_reparam(p: torch.Tensor, minimum: float) -> torch.Tensor:
    """NonNegativeParametrizer.forward: max(p, bound)**2 - pedestal."""
    bound = (minimum + PEDESTAL) ** 0.5
    ped = torch.tensor([PEDESTAL], dtype=p.dtype)
    out = torch.max(p, torch.tensor([bound], dtype=p.dtype))
    return out ** 2 - ped


def gdn_init_params(channels: int, gamma_init: float = 0.1) -> Tuple[torch.Tensor, torch.Tensor]:
    beta = nonneg_init(torch.ones(channels))
    gamma = nonneg_init(gamma_init * torch.eye(channels))
    return beta, gamma


def gdn_forward(x: torch.Tensor, beta: torch.Tensor, gamma: torch.Tensor,
                inverse: bool = False, beta_min: float = 1e-6) -> torch.Tensor:
    """x (B,C,H,W) fp32; beta (C), gamma (C,C) are the *stored* (pre-reparam) tensors."""
    C = x.shape[1]
    b = nonneg_reparam(beta, beta_min)
    g = nonneg_reparam(gamma, 0.0).reshape(C, C, 1, 1)
    norm = F.conv2d(x ** 2, g, b)
    norm = torch.sqrt(norm) if inverse else torch.rsqrt(norm)
    return x * norm


# --------------------------------------------------------------------------------------
# Conv stacks (reference-pinned)
# --------------------------------------------------------------------------------------

def reflect_conv_s2(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None,
                    stride: int = 2) -> torch.Tensor:
    """nn.Conv2d(k, stride, padding=k//2, padding_mode='reflect') -- _autoencoders.py:78-85."""
    k = w.shape[-1]
    p = k // 2
    xp = F.pad(x, (p, p, p, p), mode='reflect') if p > 0 else x
    return F.conv2d(xp, w, bias, stride=stride)


def deconv_s2(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """nn.ConvTranspose2d(k, stride 2, padding k//2, output_padding 1) -- _autoencoders.py:204-211."""
    k = w.shape[-1]
    return F.conv_transpose2d(x, w, bias, stride=2, padding=k // 2, output_padding=1)


def _act(fx: torch.Tensor, act: Optional[str]) -> torch.Tensor:
    """_define_act_layer (_autoencoders.py:19-34): nn.LeakyReLU() (slope 0.01) / nn.ReLU()."""
    if act == 'LeakyReLU':
        return F.leaky_relu(fx, 0.01)
    if act == 'ReLU':
        return F.relu(fx)
    return fx


def analysis_forward(x: torch.Tensor, layers: Sequence[dict]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Analyzer: layers[i] = {'weight', 'bias'?, 'beta'?, 'gamma'?, 'pre_weight'?, 'pre_bias'?, 'act'?}.

    GDN units (DownsamplingUnit with act_layer_type='GDN'): model.0 = conv s2, model.1 = GDN.
    LeakyReLU / ReLU units (_autoencoders.py:62-92): model.0 = conv(cin,cin,s1,reflect), model.1 = act,
    model.2 = conv s2, model.3 = act.  Last unit has act None (:343-351).  Returns (y, per-unit outputs).
    """
    outs = []
    fx = x
    for L in layers:
        if L.get('pre_weight') is not None:
            fx = _act(reflect_conv_s2(fx, L['pre_weight'], L.get('pre_bias'), stride=1), L.get('act'))
        fx = reflect_conv_s2(fx, L['weight'], L.get('bias'))
        if L.get('beta') is not None:
            fx = gdn_forward(fx, L['beta'], L['gamma'], inverse=False)
        else:
            fx = _act(fx, L.get('act'))
        outs.append(fx)
    return fx, outs


def synthesis_forward(y: torch.Tensor, layers: Sequence[dict]) -> Tuple[torch.Tensor, List[torch.Tensor]]:
    """Synthesizer (UpsamplingUnit :177-227): GDN units model.0 = convT s2, model.1 = IGDN; LeakyReLU / ReLU
    units model.0 = convT(cin,cin,s1,p=k//2), model.1 = act, model.2 = convT s2, model.3 = act."""
    outs = []
    fx = y
    for L in layers:
        if L.get('pre_weight') is not None:
            k = L['pre_weight'].shape[-1]
            fx = _act(F.conv_transpose2d(fx, L['pre_weight'], L.get('pre_bias'), stride=1, padding=k // 2),
                      L.get('act'))
        fx = deconv_s2(fx, L['weight'], L.get('bias'))
        if L.get('beta') is not None:
            fx = gdn_forward(fx, L['beta'], L['gamma'], inverse=True)
        else:
            fx = _act(fx, L.get('act'))
        outs.append(fx)
    return fx, outs


def tile_to_input(tile_u8: np.ndarray) -> torch.Tensor:
    """(h,w,c) uint8 -> (1,c,h,w) fp32 / 255  -- _autoencoders.py:542-545."""
    t = torch.from_numpy(np.ascontiguousarray(tile_u8)).permute(2, 0, 1)
    return (t.float() / 255.0).unsqueeze(0)


def output_to_tile(x_r: torch.Tensor) -> np.ndarray:
    """(c,h,w) fp32 -> (h,w,c) uint8: *255, clip, truncating cast -- _autoencoders.py:576-580."""
    t = (x_r * 255.0).clip(0, 255).to(torch.uint8)
    return np.ascontiguousarray(t.permute(1, 2, 0).numpy())


# --------------------------------------------------------------------------------------
# pmf_to_quantized_cdf (compressai._CXX, Appendix A.2)  -- pure python restatement
# --------------------------------------------------------------------------------------

def pmf_to_quantized_cdf(pmf: Sequence[float], precision: int = 16) -> List[int]:
    pmf32 = np.asarray(pmf, dtype=np.float32)
    if not np.all(np.isfinite(pmf32)) or np.any(pmf32 < 0):
        raise ValueError("Invalid `pmf`, non-finite or negative element found")
    scale = np.float32(1 << precision)
    cdf = [0] * (len(pmf32) + 1)
    for i, p in enumerate(pmf32):
        v = np.float32(p) * scale
        # std::round: half away from zero (values are >= 0)
        cdf[i + 1] = int(math.floor(float(v) + 0.5))
    total = sum(cdf) & 0xFFFFFFFF
    if total == 0:
        raise ValueError("Invalid `pmf`: at least one element must have a non-zero probability.")
    cdf = [((1 << precision) * c) // total for c in cdf]
    acc = 0
    for i in range(len(cdf)):
        acc += cdf[i]
        cdf[i] = acc
    cdf[-1] = 1 << precision
    n = len(cdf)
    for i in range(n - 1):
        if cdf[i] == cdf[i + 1]:
            best_freq = 0xFFFFFFFF
            best_steal = -1
            for j in range(n - 1):
                freq = cdf[j + 1] - cdf[j]
                if freq > 1 and freq < best_freq:
                    best_freq = freq
                    best_steal = j
            assert best_steal != -1
            if best_steal < i:
                for j in range(best_steal + 1, i + 1):
                    cdf[j] -= 1
            else:
                assert best_steal > i
                for j in range(i + 1, best_steal + 1):
                    cdf[j] += 1
    assert cdf[0] == 0 and cdf[-1] == (1 << precision)
    for i in range(n - 1):
        assert cdf[i + 1] > cdf[i]
    return cdf


# --------------------------------------------------------------------------------------
# rANS64 (compressai.ans / ryg_rans rans64.h, Appendix A.3) -- pure python restatement
# (slow: small cases only; the C restatement in rans64_oracle.c is the fast checker)
# --------------------------------------------------------------------------------------

RANS64_L = 1 << 31
PRECISION = 16
BYPASS_PRECISION = 4
MAX_BYPASS_VAL = (1 << BYPASS_PRECISION) - 1
M64 = (1 << 64) - 1


def rans_symbolize(symbols, indexes, cdfs, cdf_lengths, offsets):
    """Per-symbol (start, range, bypass) stack exactly as BufferedRansEncoder builds it."""
    syms = []
    for s, idx in zip(symbols, indexes):
        cdf = cdfs[idx]
        max_value = cdf_lengths[idx] - 2
        value = int(s) - int(offsets[idx])
        raw_val = 0
        if value < 0:
            raw_val = -2 * value - 1
            value = max_value
        elif value >= max_value:
            raw_val = 2 * (value - max_value)
            value = max_value
        if raw_val >= (1 << 28):
            raise ValueError('symbol outside the codable range (upstream shifts a uint32 by 32 here)')
        syms.append((int(cdf[value]) & 0xFFFF, (int(cdf[value + 1]) - int(cdf[value])) & 0xFFFF, False))
        if value == max_value:
            n_bypass = 0
            while (raw_val >> (n_bypass * BYPASS_PRECISION)) != 0:
                n_bypass += 1
            val = n_bypass
            while val >= MAX_BYPASS_VAL:
                syms.append((MAX_BYPASS_VAL, MAX_BYPASS_VAL + 1, True))
                val -= MAX_BYPASS_VAL
            syms.append((val, val + 1, True))
            for j in range(n_bypass):
                v = (raw_val >> (j * BYPASS_PRECISION)) & MAX_BYPASS_VAL
                syms.append((v, v + 1, True))
    return syms


def rans_encode_with_indexes(symbols, indexes, cdfs, cdf_lengths, offsets) -> bytes:
    syms = rans_symbolize(symbols, indexes, cdfs, cdf_lengths, offsets)
    x = RANS64_L
    words: List[int] = []  # emitted in reverse memory order
    for start, rng, bypass in reversed(syms):
        if not bypass:
            x_max = ((RANS64_L >> PRECISION) << 32) * rng
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x // rng) << PRECISION) + (x % rng) + start
        else:
            freq = 1 << (16 - BYPASS_PRECISION)
            x_max = ((RANS64_L >> 16) << 32) * freq
            if x >= x_max:
                words.append(x & 0xFFFFFFFF)
                x >>= 32
            x = ((x << BYPASS_PRECISION) | start) & M64
    # flush: memory order word0 = low, word1 = high, then previously emitted words reversed
    out = [x & 0xFFFFFFFF, (x >> 32) & 0xFFFFFFFF] + list(reversed(words))
    return struct.pack('<%dI' % len(out), *out)


def rans_decode_with_indexes(encoded: bytes, indexes, cdfs, cdf_lengths, offsets) -> List[int]:
    n_words = len(encoded) // 4
    words = struct.unpack('<%dI' % n_words, encoded[:4 * n_words])
    x = words[0] | (words[1] << 32)
    ptr = 2

    def get_bits(nbits):
        nonlocal x, ptr
        val = x & ((1 << nbits) - 1)
        x >>= nbits
        if x < RANS64_L:
            x = (x << 32) | words[ptr]
            ptr += 1
        return val

    out = []
    for idx in indexes:
        cdf = cdfs[idx]
        n = cdf_lengths[idx]
        max_value = n - 2
        cum = x & ((1 << PRECISION) - 1)
        k = 0
        while k < n and not (int(cdf[k]) > cum):
            k += 1
        s = k - 1
        start = int(cdf[s])
        rng = int(cdf[s + 1]) - start
        x = rng * (x >> PRECISION) + (x & ((1 << PRECISION) - 1)) - start
        if x < RANS64_L:
            x = (x << 32) | words[ptr]
            ptr += 1
        value = s
        if value == max_value:
            val = get_bits(BYPASS_PRECISION)
            n_bypass = val
            while val == MAX_BYPASS_VAL:
                val = get_bits(BYPASS_PRECISION)
                n_bypass += val
            raw_val = 0
            for j in range(n_bypass):
                val = get_bits(BYPASS_PRECISION)
                raw_val |= val << (j * BYPASS_PRECISION)
            value = raw_val >> 1
            if raw_val & 1:
                value = -value - 1
            else:
                value += max_value
        out.append(value + int(offsets[idx]))
    return out


# --------------------------------------------------------------------------------------
# Factorized EntropyBottleneck (Appendix A.2 / A.3)
# --------------------------------------------------------------------------------------

class EntropyBottleneckOracle:
    """Functional restatement of compressai.entropy_models.EntropyBottleneck.

    Parameters are held in a plain dict with the reference state-dict names
    (``_matrix{i}``, ``_bias{i}``, ``_factor{i}``, ``quantiles``;
    ``scripts/transfer_weights.py:5-19`` confirms the names).
    """

    def __init__(self, channels: int, filters: Sequence[int] = (3, 3, 3, 3),
                 init_scale: float = 10.0, tail_mass: float = 1e-9,
                 likelihood_bound: float = 1e-9, generator: Optional[torch.Generator] = None,
                 likelihood_form: Optional[str] = None):
        # 'plain' = sigmoid(u) - sigmoid(l): compressai >= 1.2.x as recalled (the reference requires >= 1.2.4);
        # 'sign_trick' = |sigmoid(s u) - sigmoid(s l)|, s = -sign(l + u): older compressai.  Unpinned either way.
        import os
        self.likelihood_form = likelihood_form or os.environ.get('CAE_LIKELIHOOD_FORM', 'plain')
        assert self.likelihood_form in ('plain', 'sign_trick')
        self.channels = int(channels)
        self.filters = tuple(int(f) for f in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.likelihood_bound = float(likelihood_bound)
        F_ = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1 / (len(self.filters) + 1))
        p = {}
        for i in range(len(self.filters) + 1):
            init = np.log(np.expm1(1 / scale / F_[i + 1]))
            p[f'_matrix{i}'] = torch.full((channels, F_[i + 1], F_[i]), float(init))
            b = torch.empty(channels, F_[i + 1], 1)
            b.uniform_(-0.5, 0.5, generator=generator)
            p[f'_bias{i}'] = b
            if i < len(self.filters):
                p[f'_factor{i}'] = torch.zeros(channels, F_[i + 1], 1)
        q = torch.tensor([-self.init_scale, 0.0, self.init_scale])
        p['quantiles'] = q.repeat(channels, 1, 1)
        self.params = p
        t = np.log(2 / self.tail_mass - 1)
        self.target = torch.tensor([-t, 0.0, t], dtype=torch.float32)
        self._offset = None
        self._quantized_cdf = None
        self._cdf_length = None

    def load(self, state: dict):
        for k in list(self.params.keys()):
            if k in state:
                self.params[k] = state[k].detach().clone().float()

    # -- density model ---------------------------------------------------------------
    def medians(self) -> torch.Tensor:
        return self.params['quantiles'][:, :, 1:2]

    def logits_cumulative(self, v: torch.Tensor) -> torch.Tensor:
        logits = v
        n = len(self.filters)
        for i in range(n + 1):
            logits = torch.matmul(F.softplus(self.params[f'_matrix{i}']), logits)
            logits = logits + self.params[f'_bias{i}']
            if i < n:
                logits = logits + torch.tanh(self.params[f'_factor{i}']) * torch.tanh(logits)
        return logits

    def likelihood(self, v: torch.Tensor):
        lower = self.logits_cumulative(v - 0.5)
        upper = self.logits_cumulative(v + 0.5)
        if self.likelihood_form == 'plain':
            lik = torch.sigmoid(upper) - torch.sigmoid(lower)
        else:
            sign = -torch.sign(lower + upper)
            lik = torch.abs(torch.sigmoid(sign * upper) - torch.sigmoid(sign * lower))
        return lik, lower, upper

    def forward(self, x: torch.Tensor, training: bool = False, generator=None):
        """x (B,C,...) -> (y_hat, likelihood) in the input layout (eval: quantise)."""
        perm = list(range(x.dim()))
        perm[0], perm[1] = 1, 0
        xp = x.permute(*perm).contiguous()
        shape = xp.shape
        values = xp.reshape(shape[0], 1, -1)
        if training:
            noise = torch.empty_like(values).uniform_(-0.5, 0.5, generator=generator)
            outputs = values + noise
        else:
            m = self.medians()
            outputs = torch.round(values - m) + m
        lik, _, _ = self.likelihood(outputs)
        if self.likelihood_bound > 0:
            lik = torch.clamp(lik, min=self.likelihood_bound)
        outputs = outputs.reshape(shape).permute(*perm).contiguous()
        lik = lik.reshape(shape).permute(*perm).contiguous()
        return outputs, lik

    def loss(self) -> torch.Tensor:
        logits = self.logits_cumulative(self.params['quantiles'])
        return torch.abs(logits - self.target).sum()

    # -- CDF tables ------------------------------------------------------------------
    def update(self):
        q = self.params['quantiles']
        medians = q[:, 0, 1]
        minima = torch.clamp(torch.ceil(medians - q[:, 0, 0]).int(), min=0)
        maxima = torch.clamp(torch.ceil(q[:, 0, 2] - medians).int(), min=0)
        self._offset = -minima
        pmf_start = medians - minima
        pmf_length = maxima + minima + 1
        max_length = int(pmf_length.max().item())
        samples = torch.arange(max_length)
        samples = samples[None, :] + pmf_start[:, None, None]
        pmf, lower, upper = self.likelihood(samples)
        pmf = pmf[:, 0, :]
        tail = torch.sigmoid(lower[:, 0, :1]) + torch.sigmoid(-upper[:, 0, -1:])
        cdf = torch.zeros((self.channels, max_length + 2), dtype=torch.int32)
        for i in range(self.channels):
            prob = torch.cat((pmf[i, :int(pmf_length[i])], tail[i]), dim=0)
            c = pmf_to_quantized_cdf(prob.tolist(), PRECISION)
            cdf[i, :len(c)] = torch.tensor(c, dtype=torch.int64).to(torch.int32)
        self._quantized_cdf = cdf
        self._cdf_length = (pmf_length + 2).int()
        return True

    # -- coding ------------------------------------------------------------------------
    def symbols(self, x: torch.Tensor) -> torch.Tensor:
        """x (B,C,H,W) -> int32 symbols = round(x - median_c) (half-to-even)."""
        m = self.medians().reshape(1, -1, *([1] * (x.dim() - 2)))
        return torch.round(x - m).int()

    def indexes(self, size) -> torch.Tensor:
        C = size[1]
        view = [1] * len(size)
        view[1] = -1
        return torch.arange(C).view(*view).int().repeat(size[0], 1, *size[2:])

    def compress(self, x: torch.Tensor, encode_fn=None) -> List[bytes]:
        enc = encode_fn or rans_encode_with_indexes
        sym = self.symbols(x)
        idx = self.indexes(x.shape)
        cdfs = self._quantized_cdf.tolist()
        lens = self._cdf_length.tolist()
        offs = self._offset.tolist()
        return [enc(sym[b].reshape(-1).tolist(), idx[b].reshape(-1).tolist(), cdfs, lens, offs)
                for b in range(x.shape[0])]

    def decompress(self, strings: Sequence[bytes], size, decode_fn=None) -> torch.Tensor:
        dec = decode_fn or rans_decode_with_indexes
        out_size = (len(strings), self.channels, *size)
        idx = self.indexes(out_size)
        cdfs = self._quantized_cdf.tolist()
        lens = self._cdf_length.tolist()
        offs = self._offset.tolist()
        out = torch.empty(out_size, dtype=torch.int32)
        for b, s in enumerate(strings):
            vals = dec(s, idx[b].reshape(-1).tolist(), cdfs, lens, offs)
            out[b] = torch.tensor(vals, dtype=torch.int32).reshape(out_size[1:])
        m = self.medians().reshape(1, -1, *([1] * len(size)))
        return out.to(m.dtype) + m


# --------------------------------------------------------------------------------------
# Codec byte format (reference _autoencoders.py:539-584)
# --------------------------------------------------------------------------------------

def codec_encode(tile_u8: np.ndarray, enc_layers, eb: EntropyBottleneckOracle, encode_fn=None) -> bytes:
    h, w, _ = tile_u8.shape
    x = tile_to_input(tile_u8)
    y, _ = analysis_forward(x, enc_layers)
    return struct.pack('>QQ', h, w) + eb.compress(y, encode_fn)[0]


def codec_decode(buf: bytes, dec_layers, eb: EntropyBottleneckOracle, decode_fn=None) -> np.ndarray:
    L = len(dec_layers)
    h, w = struct.unpack('>QQ', buf[:16])
    size = (h // 2 ** L, w // 2 ** L)
    yq = eb.decompress([buf[16:]], size, decode_fn)
    x_r, _ = synthesis_forward(yq, dec_layers)
    return output_to_tile(x_r[0])


# --------------------------------------------------------------------------------------
# Metrics of the reference's harness (src/test_cae.py:47-73)
# --------------------------------------------------------------------------------------
def ssim_uint8(x: np.ndarray, x_r: np.ndarray) -> float:
    """skimage.metrics.structural_similarity(x, x_r, channel_axis=2) for uint8 (H, W, C) images, as called at
    test_cae.py:55-57.  skimage is absent from this image ("parity unpinned"): restated from the published
    algorithm -- float64, 7x7 uniform window (scipy.ndimage.uniform_filter), sample covariance (NP/(NP-1)),
    K1 = 0.01, K2 = 0.03, data range 255, crop (win-1)//2 = 3 pixels, mean per channel, then mean over channels."""
    from scipy.ndimage import uniform_filter
    win, k1, k2, rng = 7, 0.01, 0.03, 255.0
    npx = win * win
    cov_norm = npx / (npx - 1.0)
    c1, c2 = (k1 * rng) ** 2, (k2 * rng) ** 2
    pad = (win - 1) // 2
    vals = []
    for c in range(x.shape[2]):
        a, b = x[..., c].astype(np.float64), x_r[..., c].astype(np.float64)
        ux, uy = uniform_filter(a, size=win), uniform_filter(b, size=win)
        uxx, uyy, uxy = uniform_filter(a * a, size=win), uniform_filter(b * b, size=win), uniform_filter(a * b, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        s = ((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2))
        vals.append(s[pad:-pad, pad:-pad].mean(dtype=np.float64))
    return float(np.mean(vals))


def psnr_uint8(x: np.ndarray, x_r: np.ndarray, max_val: float = 255.0) -> float:
    """test_cae.py:60-63 computed in float64 (the reference subtracts uint8 arrays, which wraps: Appendix B)."""
    mse = np.mean((x.astype(np.float64) - x_r.astype(np.float64)) ** 2)
    return float(20 * math.log10(max_val) - 10 * math.log10(mse))


def delta_cielab_uint8(x: np.ndarray, x_r: np.ndarray) -> float:
    """np.mean(deltaE_cie76(rgb2lab(x), rgb2lab(x_r))) of test_cae.py:21-45 for uint8 RGB images.  skimage is absent
    ("parity unpinned"): restated from skimage.color.colorconv -- img_as_float, sRGB companding inverse (threshold
    0.04045), xyz_from_rgb matrix, D65 / 2-degree white (0.95047, 1, 1.08883), f(t) = cbrt(t) above 0.008856 else
    7.787 t + 16/116, L = 116 fy - 16, a = 500 (fx - fy), b = 200 (fy - fz); float64."""
    m = np.array([[0.412453, 0.357580, 0.180423], [0.212671, 0.715160, 0.072169], [0.019334, 0.119193, 0.950227]])
    white = np.array([0.95047, 1.0, 1.08883])

    def lab(img):
        arr = img.astype(np.float64) / 255.0
        arr = np.where(arr > 0.04045, np.power((arr + 0.055) / 1.055, 2.4), arr / 12.92)
        xyz = arr @ m.T / white
        f = np.where(xyz > 0.008856, np.cbrt(xyz), 7.787 * xyz + 16.0 / 116.0)
        return np.stack([116.0 * f[..., 1] - 16.0, 500.0 * (f[..., 0] - f[..., 1]), 200.0 * (f[..., 1] - f[..., 2])], -1)

    d = lab(x) - lab(x_r)
    return float(np.mean(np.sqrt((d * d).sum(-1))))


def ms_ssim_uint8(x: np.ndarray, x_r: np.ndarray) -> float:
    """pytorch_msssim.ms_ssim(x_r, x, data_range=255) of test_cae.py:47-52 for uint8 (H, W, C) images, in torch-CPU
    float32.  pytorch_msssim is absent ("parity unpinned"): restated from the published implementation -- 11-tap Gaussian
    window (sigma 1.5) applied separably without padding (rows first), K = (0.01, 0.03), five scales with weights
    (0.0448, 0.2856, 0.3001, 0.2363, 0.1333), relu of the per-channel cs / ssim means, 2x2 average pooling with
    padding s % 2 (padded zeros counted) between scales, mean over channels."""
    X = torch.from_numpy(np.moveaxis(x_r, -1, 0)[None]).float()
    Y = torch.from_numpy(np.moveaxis(x, -1, 0)[None]).float()
    C = X.shape[1]
    assert min(X.shape[-2:]) > (11 - 1) * 2 ** 4
    coords = torch.arange(11, dtype=torch.float) - 11 // 2
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    g = (g / g.sum()).view(1, 1, 1, 11).repeat(C, 1, 1, 1)
    c1, c2 = (0.01 * 255) ** 2, (0.03 * 255) ** 2

    def gauss(t):
        t = F.conv2d(t, g.transpose(2, 3), groups=C)
        return F.conv2d(t, g, groups=C)

    weights = torch.tensor([0.0448, 0.2856, 0.3001, 0.2363, 0.1333])
    mcs = []
    for lvl in range(5):
        mu1, mu2 = gauss(X), gauss(Y)
        s1, s2, s12 = gauss(X * X) - mu1 * mu1, gauss(Y * Y) - mu2 * mu2, gauss(X * Y) - mu1 * mu2
        cs_map = (2 * s12 + c2) / (s1 + s2 + c2)
        ssim_map = ((2 * mu1 * mu2 + c1) / (mu1 * mu1 + mu2 * mu2 + c1)) * cs_map
        ssim_c, cs_c = ssim_map.flatten(2).mean(-1), cs_map.flatten(2).mean(-1)
        if lvl < 4:
            mcs.append(torch.relu(cs_c))
            pad = [s % 2 for s in X.shape[2:]]
            X, Y = F.avg_pool2d(X, kernel_size=2, padding=pad), F.avg_pool2d(Y, kernel_size=2, padding=pad)
    vals = torch.stack(mcs + [torch.relu(ssim_c)], dim=0)
    return float(torch.prod(vals ** weights.view(-1, 1, 1), dim=0).mean())
