#!/usr/bin/env python3
"""ms per training step (train.train_step: forward, loss, backward, clip, Adam) of a model VARIANT at the canonical shape
(128 / 192 / L4, 256 x 256 patches): the fused track functions (plain GDN / LeakyReLU units) next to the per-operation
composition (residual, batch norm, groups).   python tools/bench_train_variant.py [batch] [steps]"""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import criteria, train

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = {}
for name, kw in (('gdn', dict(act_layer_type='GDN')),
                 ('lrelu', dict(act_layer_type='LeakyReLU')),
                 ('gdn_residual', dict(act_layer_type='GDN', use_residual=True)),
                 ('lrelu_residual', dict(act_layer_type='LeakyReLU', use_residual=True)),
                 ('lrelu_batch_norm', dict(act_layer_type='LeakyReLU', batch_norm=True))):
    torch.manual_seed(0)
    model = cae.setup_modules(channels_org=3, channels_net=128, channels_bn=192, compression_level=4, **kw)
    model = {k: m.cuda().train() for k, m in model.items()}
    crit = criteria.GeneralLoss(distortion_lambda=0.01)
    opts = train.setup_optim(model, learning_rate=1e-4)
    x = torch.rand(batch, 3, 256, 256, device='cuda')
    for _ in range(2):
        ld = train.train_step(x, model, crit, opts)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ld = train.train_step(x, model, crit, opts)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    out[name] = dict(ms_per_step=round(ms, 2), samples_per_s=round(batch / ms * 1e3, 1), loss=round(float(ld['loss']), 3))
    del model, opts
    torch.cuda.empty_cache()
print(json.dumps(dict(batch=batch, patch=256, steps=steps, **out)))
