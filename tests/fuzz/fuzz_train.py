#!/usr/bin/env python3
"""Randomised gradient-parity sweep of the TRAINING path on the GPU: random unit variants (residual, batch norm, groups, bias,
activation, kernel size, depth) and sizes; outputs and every parameter gradient of both tracks against torch-CPU autograd of
oracle/train_oracle.residual_track (the restatement with the kernels' rounding points; pinned to the reference's fixtures by
tests/test_host.py).  Canonical-style models (no residual / batch norm / groups) take the fused track functions, the others the
per-operation composition, so both are swept.
usage: fuzz_train.py [n_cases] [seed]   -> prints failures, exits 1 if any."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cnn_autoencoder_amd as cae
from conftest import residual_oracle_units
from oracle import train_oracle as T

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
print(f'fuzz_train: seed {seed}, {n_cases} cases', flush=True)


def rel(a, b):
    return float((a.double().cpu() - b.double().cpu()).abs().max() / max(float(b.abs().max()), 1e-30))


fails = tight = loose = 0
t0 = time.time()
for case in range(n_cases):
    act = rng.choice([None, 'GDN', 'LeakyReLU', 'ReLU'])
    act = None if act is None else str(act)
    groups = bool(rng.integers(0, 4) == 0)
    L = int(rng.integers(1, 4))
    if groups:  # depthwise layers need output channels divisible by the input channels
        c = int(rng.choice([4, 8]))
        enc_kw = dict(channels_org=c, channels_net=2 * c, channels_bn=4 * c)
        dec_kw = dict(channels_org=c, channels_net=c, channels_bn=c)
    else:
        enc_kw = dec_kw = dict(channels_org=int(rng.choice([1, 3])), channels_net=int(rng.choice([8, 32, 40, 64])),
                               channels_bn=int(rng.choice([16, 48, 72])))
    kw = dict(compression_level=L, kernel_size=int(rng.choice([3, 5])), bias=bool(rng.integers(0, 2)), groups=groups,
              batch_norm=bool(rng.integers(0, 3) == 0), use_residual=bool(rng.integers(0, 2)), act_layer_type=act)
    n = int(rng.integers(2, 5))
    h, w = int(rng.integers(2 ** L + 3, 49)), int(rng.integers(2 ** L + 3, 65))
    lh, lw = int(rng.integers(2, 7)), int(rng.integers(2, 9))
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    desc = f'case {case}: {kw} enc {enc_kw} {(n, h, w)} latents {(lh, lw)}'
    try:
        enc = cae.Analyzer(**enc_kw, **kw).cuda().train()
        dec = cae.Synthesizer(**dec_kw, **kw).cuda().train()
    except (ValueError, NotImplementedError) as e:
        print('skip', desc, repr(e)[:80], flush=True)
        continue
    with torch.no_grad():
        for mod in list(enc.modules()) + list(dec.modules()):
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
            if isinstance(mod, cae.GDN):
                mod.gamma.add_(0.05 * torch.rand_like(mod.gamma))
    act_name = act if act in ('LeakyReLU', 'ReLU') else None
    problems = []
    for name, mod, track, inp, synthesis in (
            ('analysis', enc, enc.analysis_track, torch.rand(n, enc_kw['channels_org'], h, w), False),
            ('synthesis', dec, dec.synthesis_track, 2.0 * torch.randn(n, dec_kw['channels_bn'], lh, lw), True)):
        units, pairs = residual_oracle_units(track, act_name)
        xin = inp.clone().requires_grad_(True)
        ref = T.residual_track(xin, units, synthesis, bf16=True)
        if not bool(torch.isfinite(ref).all()) or float(ref.detach().abs().max()) > 1e4:
            continue  # an untrained residual / IGDN stack can blow up in the restatement itself
        # the same restatement WITHOUT the bf16 rounding points: how far the roundings alone move each gradient.  Untrained
        # batch-norm / residual stacks are ill-conditioned (a 3-level k = 5 LeakyReLU model: 4 - 40 % per parameter), and two
        # summation orders of the same roundings then differ by about as much (measured 0.5 - 1.5 x) -- the allowance is
        # max(1e-2, 2 x that sensitivity): well-conditioned gradients are held to 1e-2, the count of those is printed.
        units32, pairs32 = residual_oracle_units(track, act_name)
        xin32 = inp.clone().requires_grad_(True)
        ref32 = T.residual_track(xin32, units32, synthesis, bf16=False)
        xdev = inp.cuda().requires_grad_(True)
        out = mod(xdev)
        out = out[0][0] if synthesis else out
        g = torch.randn_like(ref.detach())
        ref.backward(g)
        ref32.backward(g)
        out.backward(g.cuda())
        scale = max(1.0, float(ref.detach().abs().max()))
        e_out = float((out.detach().cpu() - ref.detach()).abs().max()) / scale
        s_out = float((ref32.detach() - ref.detach()).abs().max()) / scale
        if e_out > max(3e-3, 2.0 * s_out):
            problems.append(f'{name} output {e_out:.2e} (rounding sensitivity {s_out:.2e})')
        sens = {n_: rel(l.grad, l32.grad) for (n_, l), (_, l32) in zip(pairs, pairs32)}
        # ... and how far an input perturbation of the size of one bf16 rounding (2^-9 relative) moves them: a gradient that is a
        # small difference of large sums (the batch-norm bias of a 1 -> 1 stage: 17 % under such a perturbation while the
        # bf16-vs-fp32 gap of the same restatement showed 0.2 %) is recognised only this way
        for _ in range(2):
            units_p, pairs_p = residual_oracle_units(track, act_name)
            xp = (inp * (1 + 2.0 ** -9 * torch.randn_like(inp))).requires_grad_(True)
            T.residual_track(xp, units_p, synthesis, bf16=True).backward(g)
            for (n_, l), (_, lp) in zip(pairs, pairs_p):
                sens[n_] = max(sens[n_], rel(lp.grad, l.grad))
        got = {k: p.grad.detach().cpu() for k, p in mod.named_parameters() if p.grad is not None}
        prefix = 'synthesis_track.' if synthesis else 'analysis_track.'
        if len(got) != len(pairs):
            problems.append(f'{name}: {len(got)} gradients for {len(pairs)} parameters')
            continue
        gmax = max(float(leaf.grad.abs().max()) for _, leaf in pairs)
        for pname, leaf in pairs:
            mine = got[prefix + pname]
            if float(leaf.grad.abs().max()) < 3e-3 * gmax:  # structurally zero gradients hold rounding noise on both sides
                ok = float(mine.abs().max()) < 2e-2 * gmax
                e = float(mine.abs().max()) / gmax
            else:
                e = rel(mine, leaf.grad)
                ok = e < max(1e-2, 2.0 * sens[pname])
                tight += sens[pname] < 4e-3
                loose += sens[pname] >= 4e-3
            if not ok:
                problems.append(f'{name} {pname} {e:.2e} (rounding sensitivity {sens[pname]:.2e})')
        if synthesis:
            e = rel(xdev.grad, xin.grad)
            if e > max(1e-2, 2.0 * rel(xin.grad, xin32.grad)):
                problems.append(f'synthesis latent gradient {e:.2e}')
    if problems:
        fails += 1
        print('FAIL', desc, problems, flush=True)
    if (case + 1) % 10 == 0:
        print(f'... {case + 1} cases, {fails} failures so far, {time.time() - t0:.0f} s', flush=True)
print(f'{n_cases} cases, {fails} failures, {time.time() - t0:.0f} s; {tight} gradients held to 1e-2, {loose} to twice their rounding sensitivity')
sys.exit(1 if fails else 0)
