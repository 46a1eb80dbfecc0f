#!/usr/bin/env python3
"""Host-only replay of tests/fuzz/fuzz_parity.py's case generator (the parameters of a case are drawn before anything
touches the GPU): names the cases of a sweep from its seed, or finds the seed of a sweep from cases its log printed.
    replay_fuzz.py list SEED FIRST LAST          -> one line per case
    replay_fuzz.py find 'CASE:key=value,...' ... -> seeds in [0, 2000) whose sweep has those cases
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cnn_autoencoder_amd as cae  # noqa: E402  (module constructors only: no device needed)

CONSTRUCT = '--construct' in sys.argv
if CONSTRUCT:
    sys.argv.remove('--construct')
FIRST = 0


def cases(seed, last):
    """the generator of fuzz_parity.py, draw for draw"""
    rng = np.random.default_rng(seed)
    for case in range(last + 1):
        L = int(rng.integers(1, 5))
        ks = int(rng.choice([3, 5]))
        act = rng.choice([None, 'GDN', 'LeakyReLU', 'ReLU'])
        act = None if act is None else str(act)
        kw = dict(channels_org=int(rng.choice([1, 3, 4])), channels_net=int(rng.choice([4, 8, 24, 32, 40, 64, 96, 128, 160])),
                  channels_bn=int(rng.choice([4, 16, 48, 72, 192])), compression_level=L, channels_expansion=1, kernel_size=ks,
                  groups=False, batch_norm=bool(rng.integers(0, 2)), dropout=0.0, bias=bool(rng.integers(0, 2)),
                  use_residual=bool(rng.integers(0, 2)), act_layer_type=act,
                  multiscale_analysis=bool(rng.integers(0, 4) == 0))
        if rng.integers(0, 4) == 0 and kw['channels_net'] <= 16 and not kw['multiscale_analysis']:
            kw['channels_expansion'] = 2
        min_side = 2 ** L + 1
        h, w = int(rng.integers(min_side, 80)), int(rng.integers(min_side, 120))
        n = int(rng.integers(1, 4))
        rng.integers(0, 1 << 30)  # (torch.manual_seed)
        # the constructors refuse none of the drawn variants (modules._check_variant: kernel 3 / 5, known activations,
        # expansion >= 1), so the sweep draws on; `--construct` checks that on the listed cases
        if CONSTRUCT and case >= FIRST:
            cae.Analyzer(**kw), cae.Synthesizer(**kw)
        prec = str(rng.choice(['fp32', 'f16x3']))
        rng.integers(0, 256, (n, h, w, kw['channels_org']), dtype=np.uint8)
        yield case, prec, kw, (n, h, w)


def main():
    if sys.argv[1] == 'list':
        global FIRST
        seed, first, last = (int(v) for v in sys.argv[2:5])
        FIRST = first
        for case, prec, kw, shape in cases(seed, last):
            if case >= first:
                print(case, prec, kw, shape)
    else:
        want = {}
        for spec in sys.argv[2:]:
            c, kv = spec.split(':')
            want[int(c)] = {k: v for k, v in (p.split('=') for p in kv.split(','))}
        for seed in range(2000):
            ok = True
            for case, prec, kw, shape in cases(seed, max(want)):
                if case in want:
                    d = dict(kw, prec=prec, n=shape[0], h=shape[1], w=shape[2])
                    if any(str(d[k]) != v for k, v in want[case].items()):
                        ok = False
                        break
            if ok:
                print('seed', seed)


if __name__ == '__main__':
    main()
