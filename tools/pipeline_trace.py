#!/usr/bin/env python3
"""Timeline of SlideCoder.run: GPU events of every A (analysis) / D (synthesis) stage and host coder spans."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import slide, synth

B, K = 32, 10
state = synth.synthetic_state(dict(synth.CANONICAL), seed=0)
codec = cae.ConvolutionalAutoencoder(checkpoint=state)
eb = codec._model['fact_ent'].module
eb.fit_quantiles(); eb.update(force=True)
coder = slide.SlideCoder(codec)
tiles = torch.from_numpy(np.concatenate([synth.histo_tiles(2, 1024)] * (B // 2))).cuda()
coder.run([tiles] * 3)

# monkeypatch stage functions with event brackets
ev = []
orig_enc, orig_dec = coder.enc.forward_u8, coder.dec.forward_u8
def wrap(fn, label):
    def f(x):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); r = fn(x); b.record()
        ev.append((label, a, b)); return r
    return f
coder.enc.forward_u8 = wrap(orig_enc, 'A')
coder.dec.forward_u8 = wrap(orig_dec, 'D')
host = []
oe, od = eb.encode_symbols, eb.decode_symbols
def we(*a, **k):
    t = time.perf_counter(); r = oe(*a, **k); host.append(('enc', t, time.perf_counter())); return r
def wd(*a, **k):
    t = time.perf_counter(); r = od(*a, **k); host.append(('dec', t, time.perf_counter())); return r
eb.encode_symbols, eb.decode_symbols = we, wd
torch.cuda.synchronize()
ref = torch.cuda.Event(enable_timing=True); ref.record(); t_ref = time.perf_counter()
coder.run([tiles] * K)
torch.cuda.synchronize()
wall = time.perf_counter() - t_ref
print(f'wall {wall*1e3:.1f} ms for {K} steps -> {K*B/wall:.0f} tiles/s')
rows = [(ref.elapsed_time(a), ref.elapsed_time(b), l) for l, a, b in ev]
rows.sort()
busy = sum(b - a for a, b, _ in rows)
print('GPU A+D busy %.1f ms' % busy)
for a, b, l in rows: print(f'  GPU {l} {a:7.2f} .. {b:7.2f}  ({b-a:.2f})')
for l, a, b in host: print(f'  host {l} {(a-t_ref)*1e3:7.2f} .. {(b-t_ref)*1e3:7.2f}  ({(b-a)*1e3:.2f})')
