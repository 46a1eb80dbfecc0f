import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import cae_oracle as O
from cnn_autoencoder_amd import metrics
rng = np.random.default_rng(5); fails = 0
for case in range(80):
    c = int(rng.choice([1, 3])); h, w = int(rng.integers(7, 300)), int(rng.integers(7, 300)); n = int(rng.integers(1, 4))
    x = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
    amp = int(rng.integers(0, 60))
    y = np.clip(x.astype(int) + rng.integers(-amp, amp + 1, x.shape), 0, 255).astype(np.uint8)
    xs, ys = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    try:
        np.testing.assert_allclose(metrics.compute_ssim(x=xs, x_r=ys).cpu().numpy(), [O.ssim_uint8(a, b) for a, b in zip(x, y)], rtol=1e-11)
        if amp:
            np.testing.assert_allclose(metrics.compute_psnr(x=xs, x_r=ys).cpu().numpy(), [O.psnr_uint8(a, b) for a, b in zip(x, y)], rtol=1e-12)
        if c == 3:
            np.testing.assert_allclose(metrics.compute_deltaCIELAB(x=xs, x_r=ys).cpu().numpy(), [O.delta_cielab_uint8(a, b) for a, b in zip(x, y)], rtol=1e-10, atol=1e-12)
        if min(h, w) > 160:
            np.testing.assert_allclose(metrics.compute_ms_ssim(x=xs, x_r=ys).cpu().numpy(), [O.ms_ssim_uint8(a, b) for a, b in zip(x, y)], rtol=5e-5)
    except AssertionError as e:
        fails += 1; print('FAIL', case, (n, h, w, c), amp, str(e)[:300])
print('metrics fuzz: 80 cases,', fails, 'failures')
