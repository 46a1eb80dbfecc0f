#!/usr/bin/env python3
"""Training-step throughput of the HIP training path (BASELINE config 5: rate-distortion training, bf16 convolutions,
fp32 GDN) on one GPU: samples/s of train.train_step on synthetic 256x256 patches, canonical model.
usage: bench_train.py [batch] [steps] [patch] [graph|eager]   (graph: train.GraphedTrainStep, the default)"""
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
mode = sys.argv[4] if len(sys.argv) > 4 else 'graph'
opts = train.setup_optim(model, capturable=mode == 'graph' or os.environ.get('CAE_BENCH_CAPTURABLE') == '1')
criterion = criteria.GeneralLoss(distortion_lambda=0.01)
x = torch.rand(batch, 3, patch, patch, device='cuda')
if mode == 'graph':
    step = train.GraphedTrainStep(x, model, criterion, opts, warmup=3)
else:
    if os.environ.get('CAE_BENCH_KEEPGRADS') == '1':  # eager, but with the persistent .grad buffers of the graphed step
        from cnn_autoencoder_amd.criteria import setup_forward_func
        ff = setup_forward_func()

        def step(t):
            out = ff(t, model)
            ld = criterion(inputs=t, outputs=out, net=model)
            torch.mean(ld['loss']).backward()
            if 'entropy_loss' in ld:
                torch.mean(ld['entropy_loss']).backward()
            for opt in opts.values():
                torch.nn.utils.clip_grad_norm_(opt.param_groups[0]['params'], max_norm=1.0)
                opt.step()
                opt.zero_grad(set_to_none=False)
            return ld
    else:
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
print(json.dumps(dict(mode=mode, batch=batch, patch=patch, ms_per_step=1e3 * dt, samples_per_s=batch / dt,
                      fwd_gflop_per_sample=fwd / 1e9, approx_tflops=3 * fwd * batch / dt / 1e12,
                      loss=float(ld['loss']))))
