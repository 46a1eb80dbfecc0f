"""_GdnFn on few-channel tensors against torch autograd of the restatement: elementwise input gradient and its sum"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import cnn_autoencoder_amd as cae
from cnn_autoencoder_amd import train
from oracle import train_oracle as T
torch.manual_seed(0)
for c in (1, 3, 16, 32):
    for inverse in (False, True):
        g = cae.GDN(c, inverse=inverse).cuda()
        with torch.no_grad():
            g.gamma.add_(0.05 * torch.rand_like(g.gamma))
        x = torch.randn(4, c, 29, 38)
        xd = x.cuda().requires_grad_(True)
        y = train._GdnFn.apply(xd, inverse, *train._gdn_params(g, c))
        gy = torch.randn_like(x)
        y.backward(gy.cuda())
        xr = x.clone().requires_grad_(True)
        beta = g.beta.detach().cpu().clone().requires_grad_(True)
        gamma = g.gamma.detach().cpu().clone().requires_grad_(True)
        yr = T.gdn(xr, beta, gamma, inverse)
        yr.backward(gy)
        e_y = float((y.detach().cpu() - yr.detach()).abs().max() / yr.detach().abs().max())
        e_g = float((xd.grad.cpu() - xr.grad).abs().max() / xr.grad.abs().max())
        s_got, s_ref = float(xd.grad.double().sum()), float(xr.grad.double().sum())
        print(f'c={c} inverse={inverse}: y err {e_y:.2e}  gx err {e_g:.2e}  sum gx {s_got:.4f} vs {s_ref:.4f}  '
              f'gbeta err {float((g.beta.grad.cpu() - beta.grad).abs().max() / beta.grad.abs().max()):.2e}', flush=True)
