#!/usr/bin/env python3
"""Training-step throughput of the HIP training path (BASELINE config 5: rate-distortion training, bf16 convolutions,
fp32 GDN) on one GPU: samples/s of train.train_step on synthetic 256x256 patches, canonical model.
usage: bench_train.py [batch] [steps] [patch]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import criteria, synth, train

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
patch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
model = cae.autoencoder_from_state_dict(synth.synthetic_state(dict(synth.CANONICAL), seed=0), train=True)
opts = train.setup_optim(model)
criterion = criteria.GeneralLoss(distortion_lambda=0.01)
x = torch.rand(batch, 3, patch, patch, device='cuda')
step = lambda t: train.train_step(t, model, criterion, opts)  # noqa: E731
for _ in range(3):
    step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ld = step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
# algorithmic FLOP of one sample: forward conv + GDN both tracks, backward = 2x the convolutions (data + weight
# gradients) and 2x the GDN contractions (+1 recompute)
from bench import layer_flops
enc, dec = layer_flops(dict(synth.CANONICAL), patch, patch)
fwd = sum(enc) + sum(dec)
print(json.dumps(dict(batch=batch, patch=patch, ms_per_step=1e3 * dt, samples_per_s=batch / dt,
                      fwd_gflop_per_sample=fwd / 1e9, approx_tflops=3 * fwd * batch / dt / 1e12,
                      loss=float(ld['loss']))))
