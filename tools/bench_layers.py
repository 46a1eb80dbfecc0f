#!/usr/bin/env python3
"""Per-layer kernel times of the analysis / synthesis tracks (HIP events inside the library)."""
import argparse, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import synth
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import layer_flops

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--tile', type=int, default=1024)
ap.add_argument('--iters', type=int, default=5)
ap.add_argument('--act', default='GDN', help="'GDN' or 'none' (conv/deconv kernels without the fused normalisation)")
args = ap.parse_args()
cfg = dict(synth.CANONICAL, act_layer_type=None if args.act == 'none' else args.act)
model = cae.autoencoder_from_state_dict(synth.synthetic_state(cfg, seed=0))
enc, dec = model['encoder'].module, model['decoder'].module
tiles = torch.from_numpy(synth.uniform_tiles(args.batch, args.tile)).cuda()
y = enc.forward_u8(tiles)
yq = torch.round(y)
dec.forward_u8(yq)
enc.set_profiling(True); dec.set_profiling(True)
enc.get_profile(); dec.get_profile()
for _ in range(args.iters):
    enc.forward_u8(tiles)
    dec.forward_u8(yq)
torch.cuda.synchronize()
ems, ec = enc.get_profile(); dms, dc = dec.get_profile()
ef, df = layer_flops(cfg, args.tile, args.tile)
tot = 0
for i, f in enumerate(ef):
    ms = ems[1 + i] / ec; tot += ms
    print(f'analysis.{i}  {ms:8.3f} ms  {f * args.batch / ms / 1e9:7.1f} TFLOP/s')
for i, f in enumerate(df):
    ms = dms[1 + i] / dc; tot += ms
    print(f'synthesis.{i} {ms:8.3f} ms  {f * args.batch / ms / 1e9:7.1f} TFLOP/s')
print(f'conv-prep {ems[0]/ec:.3f} {dms[0]/dc:.3f} ms; total main kernels {tot:.3f} ms -> {args.batch / tot * 1e3:.0f} tiles/s GPU-only')
