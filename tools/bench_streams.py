#!/usr/bin/env python3
"""One-way throughput of the pipelined streams (what compress.py / decompress.py do): SlideCoder.compress_batches
(tiles -> payloads) and decompress_batches (payloads -> tiles), device-resident and host-resident input."""
import argparse, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import slide, synth

ap = argparse.ArgumentParser()
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--tile', type=int, default=1024)
ap.add_argument('--steps', type=int, default=48)
args = ap.parse_args()
state = synth.synthetic_state(dict(synth.CANONICAL), seed=0)
codec = cae.ConvolutionalAutoencoder(checkpoint=state)
eb = codec._model['fact_ent'].module
eb.fit_quantiles(); eb.update(force=True)
coder = slide.SlideCoder(codec)
host = np.concatenate([synth.histo_tiles(4, args.tile)] * (args.batch // 4))
dev = torch.from_numpy(host).cuda()
payloads = list(coder.compress_batches([dev] * 3))[0]
list(coder.decompress_batches([payloads] * 3, args.tile, args.tile))
list(coder.compress_batches([host] * 3))
for name, fn in (('compress, tiles resident in HBM', lambda: sum(len(p) for p in coder.compress_batches(dev for _ in range(args.steps)))),
                 ('compress, tiles in host memory (pinned staging + H2D)', lambda: sum(len(p) for p in coder.compress_batches(host for _ in range(args.steps)))),
                 ('decompress, reconstructions left in HBM', lambda: sum(r.size(0) for r in coder.decompress_batches((payloads for _ in range(args.steps)), args.tile, args.tile))),
                 ('decompress, reconstructions delivered in pinned host memory', lambda: sum(r.shape[0] for r in coder.decompress_batches((payloads for _ in range(args.steps)), args.tile, args.tile, to_host=True)))):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f'{name}: {args.steps * args.batch / dt:8.0f} tiles/s  ({dt / args.steps * 1e3:.2f} ms per {args.batch}-tile batch)')
