#!/usr/bin/env python3
"""GPU time of the analysis (+fused quantiser) and the synthesis (+fused dequantiser) against the batch size:
what a micro-batch of the codec front door costs.  python tools/bench_batch_sweep.py [--tile 1024]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnn_autoencoder_amd as cae  # noqa: E402
from cnn_autoencoder_amd import synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument('--tile', type=int, default=1024)
args = ap.parse_args()
state = synth.synthetic_state(dict(synth.CANONICAL), seed=0)
codec = cae.ConvolutionalAutoencoder(checkpoint=state)
enc, dec, eb = (codec._model[k].module for k in ('encoder', 'decoder', 'fact_ent'))
tiles = torch.from_numpy(synth.histo_tiles(32, args.tile)).cuda()
out = {}
for n in (1, 2, 4, 8, 16, 32):
    x = tiles[:n].contiguous()
    sym = enc.forward_u8_symbols(x, eb)
    rec = dec.forward_symbols_u8(sym, eb)
    torch.cuda.synchronize()
    res = {}
    for name, fn in (('analysis', lambda: enc.forward_u8_symbols(x, eb, defer=True)),
                     ('synthesis', lambda: dec.forward_symbols_u8(sym, eb, defer=True))):
        reps = max(4, 64 // n)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + '_ms'] = round(e0.elapsed_time(e1) / reps, 4)
        res[name + '_ms_per_tile'] = round(e0.elapsed_time(e1) / reps / n, 4)
    out[str(n)] = res
print(json.dumps(out))
