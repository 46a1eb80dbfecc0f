"""compressai-zoo <-> CAE checkpoint key mapping (scripts/transfer_weights.py of the reference)."""
import numpy as np
import pytest
import torch

from cnn_autoencoder_amd import transfer


def _zoo_like(N=16, M=24, k=5, old_names=True):
    """State dict with the key layout of compressai's bmshj2018-factorized (g_a / g_s / entropy_bottleneck)."""
    g = torch.Generator().manual_seed(0)
    sd = {}
    cin = 3
    for i in range(4):
        cout = N if i < 3 else M
        sd[f'g_a.{2 * i}.weight'] = torch.randn(cout, cin, k, k, generator=g) * 0.05
        sd[f'g_a.{2 * i}.bias'] = torch.randn(cout, generator=g) * 0.01
        if i < 3:
            sd[f'g_a.{2 * i + 1}.beta'] = torch.rand(cout, generator=g) + 0.5
            sd[f'g_a.{2 * i + 1}.gamma'] = 0.3 * torch.eye(cout) + 0.01 * torch.rand(cout, cout, generator=g)
        cin = cout
    cin = M
    for i in range(4):
        cout = N if i < 3 else 3
        sd[f'g_s.{2 * i}.weight'] = torch.randn(cin, cout, k, k, generator=g) * 0.05
        sd[f'g_s.{2 * i}.bias'] = torch.randn(cout, generator=g) * 0.01
        if i < 3:
            sd[f'g_s.{2 * i + 1}.beta'] = torch.rand(cout, generator=g) + 0.5
            sd[f'g_s.{2 * i + 1}.gamma'] = 0.3 * torch.eye(cout) + 0.01 * torch.rand(cout, cout, generator=g)
        cin = cout
    filters = (1, 3, 3, 3, 3, 1)
    for i in range(5):
        m, b, f = ('_matrices.%d', '_biases.%d', '_factors.%d') if old_names else ('_matrix%d', '_bias%d', '_factor%d')
        sd['entropy_bottleneck.' + m % i] = torch.randn(M, filters[i + 1], filters[i], generator=g)
        sd['entropy_bottleneck.' + b % i] = torch.rand(M, filters[i + 1], 1, generator=g) - 0.5
        if i < 4:
            sd['entropy_bottleneck.' + f % i] = torch.zeros(M, filters[i + 1], 1)
    sd['entropy_bottleneck.quantiles'] = torch.tensor([-10.0, 0.0, 10.0]).repeat(M, 1, 1)
    return sd


@pytest.mark.parametrize('old_names', [True, False])
def test_key_mapping_round_trip(old_names):
    src = _zoo_like(old_names=old_names)
    cae_sd = transfer.compressai_to_cae(src)
    assert 'analysis_track.0.model.0.weight' in cae_sd['encoder'] and 'analysis_track.0.model.1.gamma' in cae_sd['encoder']
    assert 'analysis_track.3.model.0.bias' in cae_sd['encoder'] and 'synthesis_track.2.model.1.beta' in cae_sd['decoder']
    assert set(cae_sd['fact_ent']) == {f'_matrix{i}' for i in range(5)} | {f'_bias{i}' for i in range(5)} | \
        {f'_factor{i}' for i in range(4)} | {'quantiles'}
    back = transfer.cae_to_compressai(cae_sd)
    new_names = _zoo_like(old_names=False)
    assert set(back) == set(new_names)
    for k in back:
        assert torch.equal(back[k], new_names[k])
    cfg = transfer.factorized_config(src)
    assert (cfg['channels_net'], cfg['channels_bn'], cfg['kernel_size'], cfg['bias'], cfg['compression_level']) == \
        (16, 24, 5, True, 4)
    assert (cfg['K'], cfg['r']) == (4, 3)


def test_transferred_checkpoint_loads_into_the_modules():
    import cnn_autoencoder_amd as cae
    src = _zoo_like()
    ckpt = dict(transfer.factorized_config(src), **transfer.compressai_to_cae(src))
    # the stored GDN tensors of a zoo file are the reparametrised ones: keys / shapes are what matters here
    model = cae.autoencoder_from_state_dict(ckpt)
    enc = model['encoder'].module
    assert torch.equal(enc.analysis_track[1].model[0].weight, src['g_a.2.weight'])
    assert torch.equal(model['decoder'].module.synthesis_track[3].model[0].bias, src['g_s.6.bias'])
    assert torch.equal(model['fact_ent'].module._matrix2, src['entropy_bottleneck._matrices.2'])
    assert model['fact_ent'].module._quantized_cdf.shape[0] == 24


@pytest.mark.gpu
def test_transferred_k5_bias_model_runs_and_matches_oracle():
    import cnn_autoencoder_amd as cae
    from cnn_autoencoder_amd import synth
    from oracle import cae_oracle as O
    from conftest import oracle_layers
    src = _zoo_like()
    for k in list(src):  # stored GDN parameters must be valid reparametrised values
        if k.endswith('.beta') or k.endswith('.gamma'):
            src[k] = torch.sqrt(src[k] + 2.0 ** -36)
    ckpt = dict(transfer.factorized_config(src), **transfer.compressai_to_cae(src))
    model = cae.autoencoder_from_state_dict(ckpt)
    tile = synth.histo_tile(64, 9, 80)
    y = model['encoder'].module.forward_u8(torch.from_numpy(tile)[None].cuda()).cpu()
    y_ref, _ = O.analysis_forward(O.tile_to_input(tile), oracle_layers(ckpt, 'encoder'))
    np.testing.assert_allclose(y.numpy(), y_ref.numpy(), rtol=1e-4, atol=1e-4)
    x_r, _ = model['decoder'](torch.round(y).cuda())
    x_ref, _ = O.synthesis_forward(torch.round(y), oracle_layers(ckpt, 'decoder'))
    np.testing.assert_allclose(x_r[0].cpu().numpy(), x_ref.numpy(), rtol=1e-4, atol=1e-4)
