import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cnn_autoencoder_amd import synth
import cnn_autoencoder_amd as cae
from oracle import cae_oracle as O
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests'))
from conftest import oracle_layers
state = synth.synthetic_state(dict(synth.CANONICAL), seed=0)
eb = cae.EntropyBottleneck(192, filters=[3]*4)
eb.load_state_dict(state['fact_ent'], strict=False)
eb.fit_quantiles(); eb.update(force=True)
enc_l = oracle_layers(state, 'encoder')
syms = []
torch.set_num_threads(8)
for i in range(4):
    t = synth.histo_tile(1024, i)
    with torch.no_grad():
        y, _ = O.analysis_forward(O.tile_to_input(t), enc_l)
    m = eb.quantiles[:, 0, 1].detach().view(1, -1, 1, 1)
    s = torch.round(y - m).int().numpy()[0]
    syms.append(s); print(i, s.min(), s.max(), flush=True)
np.savez('gpurun_out/syms.npz', sym=np.stack(syms), cdf=eb._quantized_cdf.numpy(), length=eb._cdf_length.numpy(),
         offset=eb._offset.numpy(), medians=eb.quantiles[:, 0, 1].detach().numpy())
