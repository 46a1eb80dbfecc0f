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
