"""Debug aid: residual units, f16x3 vs fp32 kernels on the same random model (per act type / track / zeroed stage)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnn_autoencoder_amd as cae

def run(mod, x):
    outs = {}
    for prec in ('fp32', 'f16x3'):
        mod.precision = prec
        y = mod(x)
        if isinstance(y, tuple):
            y = y[0][0]
        outs[prec] = y.float().cpu()
    return outs

for corg in (3, 8):
    for zero_stage in (True, False):
        torch.manual_seed(0)
        a = cae.Analyzer(corg, 8, 16, 1, use_residual=True, act_layer_type=None).cuda().eval()
        if zero_stage:
            with torch.no_grad():
                for u in a.analysis_track:
                    for mod_ in u.res_model:
                        if hasattr(mod_, 'weight'):
                            mod_.weight.zero_()
        x = torch.rand(1, corg, 32, 32).cuda()
        o = run(a, x)
        f, h = o['fp32'], o['f16x3']
        print(f'corg={corg} zero_stage={zero_stage}: |fp32| {f.abs().max():.3f} |f16| {h.abs().max():.3f} err {(f-h).abs().max():.3e}'
              f' ratio f16/fp32 median {(h / f).median():.3f}', flush=True)
        print('  fp32', f[0, 0, 0, :6].tolist())
        print('  f16 ', h[0, 0, 0, :6].tolist())
