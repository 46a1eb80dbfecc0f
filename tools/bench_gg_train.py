#!/usr/bin/env python3
"""Times the training convolution kernels alone (cae_t_conv_forward = strided gather-GEMM, cae_t_deconv_forward = its four
parity launches, cae_t_wgrad) on the canonical 128 -> 128 layer: ms and bf16 TFLOP/s.  python tools/bench_gg_train.py [batch]"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cnn_autoencoder_amd import _lib, train  # noqa: E402
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 128
L = _lib.lib()
C, ks = 128, 3
out = {}
for hw in (128, 64):
    x16 = torch.randn(batch, hw, hw, C, device='cuda').bfloat16()
    w = torch.randn(C, C, ks, ks, device='cuda') * 0.05
    wp = train._pack(w, 1, ks)
    wpt = train._pack(w, 0, ks)
    z16 = torch.empty(batch, hw // 2, hw // 2, C, device='cuda', dtype=torch.bfloat16)
    u16 = torch.empty(batch, 2 * hw, 2 * hw, C, device='cuda', dtype=torch.bfloat16) if hw == 64 else None
    gw = torch.empty(ks * ks, C, C, device='cuda')
    g16 = torch.randn(batch, hw // 2, hw // 2, C, device='cuda').bfloat16()
    calls = {'conv_fwd': lambda: _lib.check(L.cae_t_conv_forward(x16.data_ptr(), batch, hw, hw, C, wp.data_ptr(), ks, None,
                                                                  z16.data_ptr(), C, None, None)),
             'wgrad': lambda: _lib.check(L.cae_t_wgrad(x16.data_ptr(), batch, hw, hw, C, g16.data_ptr(), hw // 2, hw // 2, C, ks, 1,
                                                       gw.data_ptr(), None))}
    if u16 is not None:
        calls['deconv_fwd'] = lambda: _lib.check(L.cae_t_deconv_forward(x16.data_ptr(), batch, hw, hw, C, wpt.data_ptr(), ks, None,
                                                                         u16.data_ptr(), C, None, None))
    for name, fn in calls.items():
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        pos = batch * (hw // 2) ** 2 if name != 'deconv_fwd' else batch * hw * hw
        flop = 2.0 * pos * C * C * ks * ks
        out[f'{name}_{hw}'] = dict(ms=round(ms, 4), tflops=round(flop / ms / 1e9, 1))
print(json.dumps(out))
