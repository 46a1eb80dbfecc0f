"""GPU metrics of the reference's test harness (src/test_cae.py:47-89) on batches of uint8 tiles resident in HBM.

``metric_fun`` mirrors the reference's table of the same name for the metrics built here: 'dist' (RMSE), 'psnr', 'ssim',
'delta_cielab' and 'rate' (bits per pixel).  Each takes ``x`` / ``x_r`` as (n, h, w, c) uint8 CUDA tensors and returns a float64 CUDA
tensor of per-tile values (the reference loops over one image at a time on the host).  'delta_cielab' is skimage's rgb2lab + deltaE_cie76
restated; 'ms-ssim' (pytorch_msssim) is not built.
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


def compute_rate(x=None, nbytes=None, **kwargs) -> torch.Tensor:
    """test_cae.py:71-73: 8 * stored bytes / pixels, per tile (nbytes: stored chunk sizes)."""
    px = float(x.size(1) * x.size(2))
    return 8.0 * torch.as_tensor(list(nbytes), dtype=torch.float64, device=x.device) / px


def _not_built(name):
    def f(*args, **kwargs):
        raise NotImplementedError(f'{name} is not built')
    return f


metric_fun = {'dist': compute_rmse, 'rate': compute_rate, 'ssim': compute_ssim, 'psnr': compute_psnr,
              'ms-ssim': _not_built('ms-ssim (pytorch_msssim)'), 'delta_cielab': compute_deltaCIELAB}
