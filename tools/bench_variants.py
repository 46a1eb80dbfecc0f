#!/usr/bin/env python3
"""Analysis + synthesis time of a model variant on both arithmetic paths (HIP events around the calls):
    python tools/bench_variants.py [--act LeakyReLU|ReLU|GDN|none] [--batch 32] [--tile 1024] [--residual] [--multiscale]
(--residual / --multiscale: modules built directly with their default initialisation, the canonical 128 / 192 / L4 shape)"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument('--act', default='LeakyReLU')
ap.add_argument('--batch', type=int, default=32)
ap.add_argument('--tile', type=int, default=1024)
ap.add_argument('--residual', action='store_true')
ap.add_argument('--multiscale', action='store_true')
args = ap.parse_args()
cfg = dict(synth.CANONICAL, act_layer_type=None if args.act == 'none' else args.act)
state = synth.synthetic_state(cfg, seed=0)
tiles = torch.from_numpy(synth.uniform_tiles(args.batch, args.tile)).cuda()
out = dict(act=args.act, batch=args.batch, tile=args.tile)
out.update(residual=args.residual, multiscale=args.multiscale)
for prec in ('fp32', 'f16x3'):
    if args.residual or args.multiscale:
        kw = dict(channels_org=3, channels_net=cfg['channels_net'], channels_bn=cfg['channels_bn'],
                  compression_level=cfg['compression_level'], act_layer_type=cfg['act_layer_type'],
                  use_residual=args.residual, multiscale_analysis=args.multiscale)
        torch.manual_seed(0)
        enc, dec = cae.Analyzer(**kw).cuda().eval(), cae.Synthesizer(**kw).cuda().eval()
        model = None
    else:
        model = cae.autoencoder_from_state_dict(state)
        enc, dec = model['encoder'].module, model['decoder'].module
    enc.precision = dec.precision = prec
    y = enc.forward_u8(tiles)
    yq = torch.round(y).clamp(-20, 20)
    syn = (lambda: dec(yq)) if args.multiscale else (lambda: dec.forward_u8(yq))  # (colour layers: the float entry point)
    syn()
    torch.cuda.synchronize()
    res = {}
    for name, fn in (('analysis', lambda: enc.forward_u8(tiles)), ('synthesis', syn)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + '_ms'] = round(e0.elapsed_time(e1) / 5, 3)
    res['precision_code'] = [enc.precision_code(), dec.precision_code()]
    res['fp32_fallbacks'] = enc.fp32_fallbacks + dec.fp32_fallbacks
    out[prec] = res
    del model
out['speedup'] = round((out['fp32']['analysis_ms'] + out['fp32']['synthesis_ms']) /
                       (out['f16x3']['analysis_ms'] + out['f16x3']['synthesis_ms']), 2)
print(json.dumps(out))
