#!/usr/bin/env python3
"""Randomised check of the pipelined drivers on the GPU: SlideCoder.run / compress_batches / decompress_batches with
random batch counts, ragged batch sizes, tile sizes and pipeline depths must reproduce the unpipelined compress() /
decompress() / tile_sse() exactly.  usage: fuzz_streams.py [n_cases] [seed]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import slide, synth

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails, t0 = 0, time.time()
for case in range(n_cases):
    L = int(rng.integers(2, 5))
    cfg = dict(synth.CANONICAL, channels_net=int(rng.choice([8, 32])), channels_bn=int(rng.choice([16, 48])),
               compression_level=L, kernel_size=int(rng.choice([3, 5])))
    codec = cae.ConvolutionalAutoencoder(checkpoint=synth.synthetic_state(cfg, seed=int(rng.integers(0, 1000))))
    coder = slide.SlideCoder(codec)
    coder.depth = int(rng.integers(1, 4))
    h, w = 2 ** L * int(rng.integers(1, 6)), 2 ** L * int(rng.integers(1, 9))
    K = int(rng.integers(1, 9))
    sizes = [int(rng.integers(1, 6)) for _ in range(K)]
    host = [rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8) for n in sizes]
    dev = [torch.from_numpy(b).cuda() for b in host]
    want_p = [coder.compress(b) for b in dev]
    want_r = [coder.decompress(p, h, w) for p in want_p]
    try:
        mixed = [d if rng.integers(0, 2) else b for d, b in zip(dev, host)]
        assert list(coder.compress_batches(iter(mixed))) == want_p, 'compress_batches'
        got_r = list(coder.decompress_batches(iter(want_p), h, w))
        assert all(torch.equal(a, b) for a, b in zip(got_r, want_r)), 'decompress_batches'
        got_h = [r.copy() for r in coder.decompress_batches(iter(want_p), h, w, to_host=True)]
        assert all(np.array_equal(a, b.cpu().numpy()) for a, b in zip(got_h, want_r)), 'decompress_batches to_host'
        stats, payloads = coder.run(dev, keep_payloads=True)
        assert payloads == want_p, 'run payloads'
        sse = torch.cat([coder.tile_sse(r, d) for r, d in zip(want_r, dev)]).cpu()
        assert torch.equal(stats[:, 1], sse), 'run sse'
        assert stats[:, 0].tolist() == [len(p) + 16 for ps in want_p for p in ps], 'run bytes'
    except AssertionError as e:
        fails += 1
        print('FAIL', case, cfg['kernel_size'], L, (h, w), sizes, coder.depth, e)
print(f'{n_cases} cases, {fails} failures, {time.time() - t0:.0f} s')
sys.exit(1 if fails else 0)
