#!/usr/bin/env python3
"""Times the fused GDN training kernels alone (cae_t_gdn_forward_save / cae_t_gdn_backward_fused) on the layer shapes of
the canonical model at batch N x 256^2: ms per launch and the fraction of the fp32 MFMA peak (forward: one C x C
contraction per pixel, backward: two).  python tools/bench_gdn_train.py [batch]   (CAE_LIB selects an alternative build)"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnn_autoencoder_amd import _lib, train  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = _lib.lib()
C = 128
out = {}
for name, hw, pad, inverse in (('gdn1_128', 128, 1, False), ('gdn2_64', 64, 1, False), ('gdn3_32', 32, 1, False),
                               ('igdn3_128', 128, 0, True)):
    torch.manual_seed(0)
    z = torch.randn(batch, hw, hw, C, device='cuda')
    beta = torch.rand(C, device='cuda') + 0.5
    gamma = (0.1 * torch.eye(C) + 0.01 * torch.rand(C, C)).cuda()
    gext = torch.randn(batch, hw + 2 * pad, hw + 2 * pad, C, device='cuda')
    y16, f = train._gdn_forward(z, beta, gamma, inverse)
    reps = 10
    res = {}
    for what in ('fwd', 'bwd'):
        fn = (lambda: train._gdn_forward(z, beta, gamma, inverse)) if what == 'fwd' else \
             (lambda: train._gdn_backward(z, gext, pad, beta, gamma, inverse, f))
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        flop = 2.0 * batch * hw * hw * C * C * (1 if what == 'fwd' else 2)
        res[what] = dict(ms=round(ms, 4), frac_fp32_mfma=round(flop / (ms * 1e-3) / 157.3e12, 3))
    out[name] = res
print(json.dumps(out))
