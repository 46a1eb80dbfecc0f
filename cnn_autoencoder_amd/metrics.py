"""GPU metrics of the reference's test harness (src/test_cae.py:47-89) on batches of uint8 tiles resident in HBM.

``metric_fun`` mirrors the reference's table of the same name for the metrics built here: 'dist' (RMSE), 'psnr', 'ssim',
'delta_cielab' and 'rate' (bits per pixel).  Each takes ``x`` / ``x_r`` as (n, h, w, c) uint8 CUDA tensors and returns a float64 CUDA
tensor of per-tile values (the reference loops over one image at a time on the host).  'delta_cielab' is skimage's rgb2lab + deltaE_cie76
restated and 'ms-ssim' is pytorch_msssim's
ms_ssim restated.
"""
from __future__ import annotations

import torch

from . import _lib


def _check_pair(x: torch.Tensor, x_r: torch.Tensor):
    dev = _lib.require_gpu()
    if x.dim() != 4 or x.shape != x_r.shape or x.dtype != torch.uint8 or x_r.dtype != torch.uint8:
        raise ValueError(f'expected two uint8 (n,h,w,c) batches of one shape, got {tuple(x.shape)} {x.dtype} and '
                         f'{tuple(x_r.shape)} {x_r.dtype}')
    return x.to(dev).contiguous(), x_r.to(dev).contiguous()


@torch.no_grad()
def tile_sse(x: torch.Tensor, x_r: torch.Tensor) -> torch.Tensor:
    """Per-tile sum of squared differences (exact integers in float64): cae_tile_sse."""
    x, x_r = _check_pair(x, x_r)
    out = torch.empty(x.size(0), dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().cae_tile_sse(x_r.data_ptr(), x.data_ptr(), x.size(0), x[0].numel(), out.data_ptr(),
                                       _lib.stream_ptr()))
    return out


def compute_rmse(x=None, x_r=None, **kwargs) -> torch.Tensor:
    """test_cae.py:66-68."""
    return torch.sqrt(tile_sse(x, x_r) / x[0].numel())


def compute_psnr(x=None, x_r=None, max_val: float = 255, **kwargs) -> torch.Tensor:
    """test_cae.py:60-63 in float64 (the reference's uint8 subtraction wraps; not reproduced)."""
    mse = tile_sse(x, x_r) / x[0].numel()
    return 20.0 * torch.log10(torch.tensor(float(max_val), dtype=torch.float64, device=mse.device)) - 10.0 * torch.log10(mse)


@torch.no_grad()
def compute_ssim(x=None, x_r=None, **kwargs) -> torch.Tensor:
    """test_cae.py:55-57: skimage's structural_similarity(x, x_r, channel_axis=2) per tile (cae_tile_ssim)."""
    x, x_r = _check_pair(x, x_r)
    n, h, w, c = x.shape
    if h < 7 or w < 7:
        raise ValueError('win_size exceeds image extent.')  # skimage's message
    ws = torch.empty(n * ((h - 6 + 31) // 32) * ((w - 6 + 31) // 32), dtype=torch.float64, device=x.device)
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().cae_tile_ssim(x.data_ptr(), x_r.data_ptr(), n, h, w, c, out.data_ptr(), ws.data_ptr(),
                                        ws.numel(), _lib.stream_ptr()))
    return out


@torch.no_grad()
def compute_deltaCIELAB(x=None, x_r=None, **kwargs) -> torch.Tensor:
    """test_cae.py:21-45: mean deltaE_cie76(rgb2lab(x), rgb2lab(x_r)) per tile (cae_tile_delta_e); RGB only."""
    x, x_r = _check_pair(x, x_r)
    n, h, w, c = x.shape
    if c != 3:
        raise ValueError(f'the input array must have size 3 along `channel_axis`, got {tuple(x.shape)}')  # skimage
    bpt = min((h * w + 255) // 256, 128)
    ws = torch.empty(n * bpt, dtype=torch.float64, device=x.device)
    out = torch.empty(n, dtype=torch.float64, device=x.device)
    _lib.check(_lib.lib().cae_tile_delta_e(x.data_ptr(), x_r.data_ptr(), n, h * w, out.data_ptr(), ws.data_ptr(),
                                           ws.numel(), _lib.stream_ptr()))
    return out


_MS_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


@torch.no_grad()
def compute_ms_ssim(x=None, x_r=None, **kwargs) -> torch.Tensor:
    """test_cae.py:47-52: pytorch_msssim.ms_ssim(x_r, x, data_range=255) per tile -- five scales, 11-tap Gaussian
    window (sigma 1.5), float32 maps (cae_msssim_level, cae_avgpool2), mean over channels."""
    x, x_r = _check_pair(x, x_r)
    n, h, w, c = x.shape
    if min(h, w) <= (11 - 1) * 2 ** 4:  # pytorch_msssim's own assertion
        raise AssertionError('Image size should be larger than %d due to the 4 downsamplings in ms-ssim' % ((11 - 1) * 2 ** 4))
    L, st = _lib.lib(), _lib.stream_ptr()
    dev = x.device
    coords = torch.arange(11, dtype=torch.float32) - 11 // 2
    g = torch.exp(-(coords ** 2) / (2 * 1.5 ** 2))
    win = (g / g.sum()).to(dev)
    planes = n * c
    a = torch.empty((planes, h, w), dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    _lib.check(L.cae_u8hwc_to_planes(x_r.data_ptr(), n, h, w, c, a.data_ptr(), st))
    _lib.check(L.cae_u8hwc_to_planes(x.data_ptr(), n, h, w, c, b.data_ptr(), st))
    vals = []
    for lvl in range(5):
        hh, ww = a.shape[1], a.shape[2]
        ws = torch.empty(2 * planes * ((hh - 10 + 31) // 32) * ((ww - 10 + 31) // 32), dtype=torch.float64, device=dev)
        out = torch.empty((planes, 2), dtype=torch.float64, device=dev)
        _lib.check(L.cae_msssim_level(a.data_ptr(), b.data_ptr(), planes, hh, ww, win.data_ptr(), out.data_ptr(),
                                      ws.data_ptr(), ws.numel(), st))
        vals.append(out)
        if lvl < 4:
            oh, ow = (hh + 2 * (hh % 2) - 2) // 2 + 1, (ww + 2 * (ww % 2) - 2) // 2 + 1
            a2 = torch.empty((planes, oh, ow), dtype=torch.float32, device=dev)
            b2 = torch.empty_like(a2)
            _lib.check(L.cae_avgpool2(a.data_ptr(), planes, hh, ww, a2.data_ptr(), st))
            _lib.check(L.cae_avgpool2(b.data_ptr(), planes, hh, ww, b2.data_ptr(), st))
            a, b = a2, b2
    # prod_l relu(cs_l)^w_l * relu(ssim_4)^w_4 per (tile, channel), mean over channels (a (5, n*c) tensor: host-side glue)
    terms = [torch.relu(v[:, 1]) for v in vals[:4]] + [torch.relu(vals[4][:, 0])]
    wts = torch.tensor(_MS_WEIGHTS, dtype=torch.float64, device=dev)
    ms = torch.prod(torch.stack(terms) ** wts.view(-1, 1), dim=0)
    return ms.view(n, c).mean(dim=1)


def compute_rate(x=None, nbytes=None, **kwargs) -> torch.Tensor:
    """test_cae.py:71-73: 8 * stored bytes / pixels, per tile (nbytes: stored chunk sizes)."""
    px = float(x.size(1) * x.size(2))
    return 8.0 * torch.as_tensor(list(nbytes), dtype=torch.float64, device=x.device) / px


def _not_built(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f'{name} is not built')
    return f


metric_fun = {'dist': compute_rmse, 'rate': compute_rate, 'ssim': compute_ssim, 'psnr': compute_psnr,
              'ms-ssim': compute_ms_ssim, 'delta_cielab': compute_deltaCIELAB}
