#!/usr/bin/env python3
"""Randomised parity sweep on the GPU: random model shapes (channels, kernel size, depth, bias, activation variant,
residual, batch norm, expansion) and tile sizes against the CPU oracle / torch-CPU replay of the folded layers.
usage: fuzz_parity.py [n_cases] [seed] [first_case]   -> prints failures, exits 1 if any.
(first_case > 0: the earlier cases only advance the generator; every case is then announced before it touches the GPU, so a
faulting one can be named from the log)"""
import os, sys, time
# every over-read becomes a deterministic fault instead of a silent read of allocator slack (set before HIP starts;
# HSA_DISABLE_FRAGMENT_ALLOCATOR=0 in the environment keeps the default allocator)
os.environ.setdefault('HSA_DISABLE_FRAGMENT_ALLOCATOR', '1')
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import synth
from test_host import cpu_track

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
first_case = int(sys.argv[3]) if len(sys.argv) > 3 else 0
print(f'fuzz_parity: seed {seed}, {n_cases} cases from case {first_case}, HSA_DISABLE_FRAGMENT_ALLOCATOR='
      f'{os.environ["HSA_DISABLE_FRAGMENT_ALLOCATOR"]} (tools/replay_fuzz.py lists a sweep\'s cases on the host)', flush=True)
fails = skipped = 0
t_start = time.time()
for case in range(n_cases):
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
        kw['channels_expansion'] = 2  # (with multiscale_analysis the reference's own colour-layer channel plan breaks)
    min_side = 2 ** L + 1
    h, w = int(rng.integers(min_side, 80)), int(rng.integers(min_side, 120))
    n = int(rng.integers(1, 4))
    torch.manual_seed(int(rng.integers(0, 1 << 30)))
    try:
        enc, dec = cae.Analyzer(**kw).eval(), cae.Synthesizer(**kw).eval()
    except (ValueError, NotImplementedError) as e:
        continue
    g = torch.Generator().manual_seed(case)
    with torch.no_grad():
        for mod in list(enc.modules()) + list(dec.modules()):
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.uniform_(0.5, 1.5, generator=g); mod.bias.uniform_(-0.2, 0.2, generator=g)
                mod.running_mean.uniform_(-0.2, 0.2, generator=g); mod.running_var.uniform_(0.5, 1.5, generator=g)
    prec = str(rng.choice(['fp32', 'f16x3']))
    enc.precision = dec.precision = prec
    tiles = rng.integers(0, 256, (n, h, w, kw['channels_org']), dtype=np.uint8)
    if case < first_case:
        continue
    if first_case:
        print('case', case, prec, kw, (n, h, w), flush=True)
    x = torch.from_numpy(tiles).permute(0, 3, 1, 2).float() / 255.0
    with torch.no_grad():
        y_ref, _ = cpu_track(enc.analysis_track, x, False)
        yq = torch.round(y_ref)
        xr_ref, _ = cpu_track(dec.synthesis_track, yq, True)
        try:
            def mark(what):  # (first_case runs: name the call a fault belongs to)
                if first_case:
                    torch.cuda.synchronize()
                    print('  ->', what, flush=True)
            enc.cuda(); dec.cuda()
            mark('enc.forward_u8')
            y = enc.forward_u8(torch.from_numpy(tiles).cuda()).cpu()
            mark('dec()')
            x_r, _ = dec(yq.cuda())
            mark('dec.forward_u8')
            u8 = dec.forward_u8(yq.cuda()).cpu()  # codec path (k = 3 GDN models: product-map form of the last two layers)
            mark('entropy model')
            # fused quantiser / dequantiser entry points == the unfused calls, bit for bit
            eb = cae.EntropyBottleneck(kw['channels_bn']).eval()
            with torch.no_grad():
                eb.quantiles[:, 0, 1] += torch.linspace(-0.45, 0.45, kw['channels_bn'])
            eb.update(force=True)
            eb.cuda()
            td = torch.from_numpy(tiles).cuda()
            mark('enc.forward_u8_symbols')
            sym = enc.forward_u8_symbols(td, eb)
            mark('quantize_symbols(enc.forward_u8)')
            assert torch.equal(sym, eb.quantize_symbols(enc.forward_u8(td))), 'fused quantiser differs'
            mark('dec.forward_symbols_u8')
            a8 = dec.forward_symbols_u8(sym, eb)
            mark('dec.forward_u8(dequantize_symbols)')
            assert torch.equal(a8, dec.forward_u8(eb.dequantize_symbols(sym))), 'fused dequantiser differs'
            mark('done')
            enc.cpu(); dec.cpu()
        except Exception as e:
            fails += 1
            print('ERROR', case, prec, kw, (n, h, w), repr(e)[:200])
            continue
    if not (torch.isfinite(y_ref).all() and torch.isfinite(xr_ref).all()):
        skipped += 1  # an untrained residual / IGDN stack can overflow fp32 in the reference itself
        continue  # (large finite magnitudes are NOT skipped: the f16x3 range guard repeats such calls on fp32)
    ey = float((y - y_ref).abs().max() / max(1.0, float(y_ref.abs().max())))
    ex = float((x_r[0].cpu() - xr_ref).abs().max() / max(1.0, float(xr_ref.abs().max())))
    assert all((t is None) != kw['multiscale_analysis'] for t in x_r[1:])  # colour layers only with multiscale_analysis
    ok = y.shape == y_ref.shape and ey < 1e-4 and ex < 1e-4
    if ok and float(xr_ref.abs().max()) < 1e3:  # uint8 epilogue: <= 1 level, rarely, against the truncated reference
        ref8 = (xr_ref * 255.0).clip(0, 255).to(torch.uint8).permute(0, 2, 3, 1)
        d8 = (u8.int() - ref8.int()).abs()
        ok = int(d8.max()) <= 1 and float((d8 > 0).float().mean()) < 5e-3
    if case % 100 == 99:
        print(f'... {case + 1} cases, {fails} failures so far, {time.time() - t_start:.0f} s', flush=True)
    if not ok:
        fails += 1
        print('FAIL', case, prec, kw, (n, h, w), 'err', ey, ex)
print(f'{n_cases} cases, {fails} failures, {skipped} skipped (reference not finite), {time.time() - t_start:.0f} s')
sys.exit(1 if fails else 0)
