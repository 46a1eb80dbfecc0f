#!/usr/bin/env python3
"""Drop-in call pattern on one GPU: threads x codec.encode / codec.decode of 1024^2 tiles (see dropin_bench.py).
    python tools/bench_dropin.py [--tile 1024] [--tiles 32] [--threads 1,8,16] [--door 0|1]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))

ap = argparse.ArgumentParser()
ap.add_argument('--tile', type=int, default=1024)
ap.add_argument('--tiles', type=int, default=32)
ap.add_argument('--threads', default='1,8,16')
ap.add_argument('--door', default='1')
ap.add_argument('--budget', type=float, default=2.0)
args = ap.parse_args()
os.environ['CAE_DOOR'] = args.door

import cnn_autoencoder_amd as cae  # noqa: E402
from cnn_autoencoder_amd import synth  # noqa: E402
import dropin_bench  # noqa: E402

state = synth.synthetic_state(dict(synth.CANONICAL), seed=0)
codec = cae.ConvolutionalAutoencoder(checkpoint=state)
eb = codec._model['fact_ent'].module
eb.fit_quantiles()
eb.update(force=True)
tiles = synth.histo_tiles(args.tiles, args.tile, first_index=0)
res = dropin_bench.measure(codec, tiles, [int(t) for t in args.threads.split(',')], args.budget)
res['door'] = args.door
print(json.dumps(res))
codec.close()
