import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLD = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def load_golden(name):
    g = np.load(os.path.join(GOLD, name + '.npz'))
    cfg = json.loads(bytes(g['cfg_json']).decode())
    return g, cfg


def golden_state(g, cfg):
    """Checkpoint-shaped dict for a golden case: stored weights if present, else rebuilt from the seed."""
    import torch
    from cnn_autoencoder_amd import synth
    seed = cfg.pop('seed')
    state = synth.synthetic_state(cfg, seed=seed)
    for part in ('encoder', 'decoder'):
        for k in list(state[part].keys()):
            key = part + '/' + k
            if key in g.files:
                assert np.array_equal(g[key], state[part][k].numpy()), key  # seed rebuild == stored weights
                state[part][k] = torch.from_numpy(g[key])
    return state


def oracle_layers(state, part):
    """[{'weight','bias','beta','gamma','pre_weight','pre_bias','act'}] for oracle.analysis_forward / synthesis_forward."""
    track = 'analysis_track' if part == 'encoder' else 'synthesis_track'
    sd = state[part]
    act = state.get('act_layer_type')
    act = act if act in ('LeakyReLU', 'ReLU') else None
    layers = []
    i = 0
    while f'{track}.{i}.model.0.weight' in sd:
        pre = f'{track}.{i}.model.2.weight' in sd  # [conv s1, act, conv s2, act]
        m = 2 if pre else 0
        layers.append(dict(weight=sd[f'{track}.{i}.model.{m}.weight'],
                           bias=sd.get(f'{track}.{i}.model.{m}.bias'),
                           beta=sd.get(f'{track}.{i}.model.1.beta'),
                           gamma=sd.get(f'{track}.{i}.model.1.gamma'),
                           pre_weight=sd[f'{track}.{i}.model.0.weight'] if pre else None,
                           pre_bias=sd.get(f'{track}.{i}.model.0.bias') if pre else None,
                           act=act if pre else None))
        i += 1
    return layers


@pytest.fixture(scope='session')
def built_lib():
    import cnn_autoencoder_amd as cae
    if not os.path.exists(cae.LIB_PATH):
        cae.build()
    return cae.LIB_PATH


def residual_oracle_units(track, act_name):
    """(oracle unit dicts with leaf clones for oracle.train_oracle.residual_track, [(module parameter name, leaf)]) of a track of
    residual units, or of plain units (their stride-1 pre-convolution as the one stage, 'residual': False); batch norms as
    'bn' entries"""
    units, pairs = [], []

    def leaf(p, name):
        if p is None:
            return None
        t = p.detach().cpu().clone().requires_grad_(True)
        pairs.append((name, t))
        return t

    def wleaf(conv, name):
        """weight leaf; a grouped layer enters the restatement through its dense block-diagonal embedding (differentiable)"""
        t = leaf(conv.weight, name)
        if conv.groups == 1:
            return t
        import torch
        g, k = conv.groups, conv.kernel_size
        cin_g, cout_g = conv.in_channels // g, conv.out_channels // g
        if conv.transposed:
            dense = [torch.cat([t[ci:ci + 1] if (ci // cin_g) == j else torch.zeros(1, cout_g, k, k) for j in range(g)], dim=1)
                     for ci in range(conv.in_channels)]
        else:
            dense = [torch.cat([t[co:co + 1] if (co // cout_g) == j else torch.zeros(1, cin_g, k, k) for j in range(g)], dim=1)
                     for co in range(conv.out_channels)]
        return torch.cat(dense, dim=0)

    def bn(seq, idx, prefix):
        if idx is None:
            return None
        m = seq[idx]
        return dict(weight=leaf(m.weight, f'{prefix}.{idx}.weight'), bias=leaf(m.bias, f'{prefix}.{idx}.bias'), eps=m.eps)
    for i, u in enumerate(track):
        stages = []
        residual = hasattr(u, '_res')
        if residual:
            for k, (ci, bi, gi) in enumerate(u._res):
                conv = u.res_model[ci]
                d = dict(weight=wleaf(conv, f'{i}.res_model.{ci}.weight'), bias=leaf(conv.bias, f'{i}.res_model.{ci}.bias'),
                         bn=bn(u.res_model, bi, f'{i}.res_model'))
                if gi is not None:
                    g = u.res_model[gi]
                    d.update(beta=leaf(g.beta, f'{i}.res_model.{gi}.beta'), gamma=leaf(g.gamma, f'{i}.res_model.{gi}.gamma'))
                else:
                    d['act'] = act_name if (u.act_code and (k == 0 or u._second_stage_act)) else None
                stages.append(d)
            post_act = act_name if (u.act_code and len(u._res) == 2) else None
        else:
            if u.pre is not None:
                pi = u.pre_index
                stages.append(dict(weight=wleaf(u.pre, f'{i}.model.{pi}.weight'), bias=leaf(u.pre.bias, f'{i}.model.{pi}.bias'),
                                   bn=bn(u.model, u.pre_bn_index, f'{i}.model'), act=act_name))
            post_act = None
        mi = u.main_index
        d = dict(stages=stages, post_act=post_act, residual=residual,
                 weight=wleaf(u.main, f'{i}.model.{mi}.weight'), bias=leaf(u.main.bias, f'{i}.model.{mi}.bias'),
                 bn=bn(u.model, u.main_bn_index, f'{i}.model'))
        if u.gdn is not None:
            d.update(beta=leaf(u.gdn.beta, f'{i}.model.{u.gdn_index}.beta'), gamma=leaf(u.gdn.gamma, f'{i}.model.{u.gdn_index}.gamma'))
        else:
            d['act'] = act_name if u.act_code else None
        units.append(d)
    return units, pairs
