"""N>1 path on CPU: two gloo ranks shard a slide's tiles and all_gather the per-tile statistics."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_tiles, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cnn_autoencoder_amd import slide
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi = slide.tile_range(rank, world, n_tiles)
    # deterministic fake per-tile records keyed by the global tile index
    idx = np.arange(lo, hi)
    local = slide.tile_stats((1000 + 7 * idx).tolist(), (0.5 * idx + 1.0).tolist(), 64 * 64 * 3)
    allst = slide.gather_stats(local)
    torch.save(dict(stats=allst, range=(lo, hi)), os.path.join(out_dir, f'rank{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('n_tiles', [8, 11, 1])
def test_two_ranks_gather_identical_slide_stats(tmp_path, n_tiles):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_tiles, str(tmp_path)), nprocs=world, join=True)
    res = [torch.load(str(tmp_path / f'rank{r}.pt'), weights_only=False) for r in range(world)]
    assert res[0]['range'][0] == 0 and res[0]['range'][1] == res[1]['range'][0] and res[1]['range'][1] == n_tiles
    assert torch.equal(res[0]['stats'], res[1]['stats'])  # every rank derives the same slide statistics
    st = res[0]['stats']
    idx = torch.arange(n_tiles, dtype=torch.float64)
    assert st.shape == (n_tiles, 3)
    assert torch.equal(st[:, 0], 1000 + 7 * idx) and torch.equal(st[:, 1], 0.5 * idx + 1.0)
    from cnn_autoencoder_amd import slide
    s = slide.slide_summary(st, 64 * 64)
    assert s['tiles'] == n_tiles and s['bytes'] == float((1000 + 7 * idx).sum())


def _zarr_worker(rank, world, port, store, codec, shape, patch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cnn_autoencoder_amd import zarrio
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    img = np.random.default_rng(3).integers(0, 256, shape, dtype=np.uint8)  # every rank holds the same slide
    zarrio.compress_image(codec, None, img, store, patch_size=patch, data_group='0/0')
    # compress_image returns behind a barrier: the store is complete on every rank
    back = zarrio.decompress_image(store, data_group='0/0')
    assert np.array_equal(back, img), f'rank {rank} reads back a different image'
    dist.destroy_process_group()


@pytest.mark.parametrize('codec,shape,patch', [('Zlib', (150, 200, 3), 64), ('None', (70, 64, 1), 32),
                                               ('Zlib', (64, 64, 3), 64)])
def test_two_ranks_write_the_store_of_one_rank(tmp_path, codec, shape, patch):
    """zarrio.compress_image under torch.distributed (compress.py:101,121-128 sharded over ranks): the ranks write
    disjoint chunk files, rank 0 the metadata; the store is byte-identical to the one a single rank writes, including
    the ragged last tile block, zero-padded edge chunks and a slide with fewer tiles than ranks."""
    from cnn_autoencoder_amd import slide, zarrio
    img = np.random.default_rng(3).integers(0, 256, shape, dtype=np.uint8)
    one = str(tmp_path / 'one.zarr')
    zarrio.compress_image(codec, None, img, one, patch_size=patch, data_group='0/0')
    two = str(tmp_path / 'two.zarr')
    world = 2
    mp.spawn(_zarr_worker, args=(world, _free_port(), two, codec, shape, patch), nprocs=world, join=True)

    def tree(root):
        out = {}
        for d, _, files in os.walk(root):
            for f in files:
                p = os.path.join(d, f)
                out[os.path.relpath(p, root)] = open(p, 'rb').read()
        return out
    a, b = tree(one), tree(two)
    assert sorted(a) == sorted(b)
    assert all(a[k] == b[k] for k in a)
    n_tiles = -(-shape[0] // patch) * -(-shape[1] // patch)
    assert len([k for k in a if not os.path.basename(k).startswith('.')]) == n_tiles
    assert slide.tile_range(0, 2, n_tiles)[1] == slide.tile_range(1, 2, n_tiles)[0]


def _cae_worker(rank, world, port, store, ckpt, shape, patch):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cnn_autoencoder_amd import zarrio
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)  # one-GPU box: both ranks on cuda:0, collective on gloo
    dist.init_process_group('gloo', rank=rank, world_size=world)
    img = np.random.default_rng(7).integers(0, 256, shape, dtype=np.uint8)
    zarrio.compress_image('CAE', ckpt, img, store, patch_size=patch, data_group='0/0', batch_tiles=2)
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_ranks_write_the_cae_store_of_one_rank(tmp_path):
    """The same with the 'cae' codec on the HIP path (two ranks sharing the one GPU of the test box): chunk
    bitstreams, metadata and the decoded image equal the single-rank store; 7 tiles over 2 ranks (ragged), edge
    chunks padded to the full patch."""
    from cnn_autoencoder_amd import synth, zarrio
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    ckpt = str(tmp_path / 'ckpt.pth')
    torch.save(synth.synthetic_state(cfg, seed=4), ckpt)
    shape, patch = (200, 150, 3), 64
    img = np.random.default_rng(7).integers(0, 256, shape, dtype=np.uint8)
    one = str(tmp_path / 'one.zarr')
    zarrio.compress_image('CAE', ckpt, img, one, patch_size=patch, data_group='0/0', batch_tiles=2)
    two = str(tmp_path / 'two.zarr')
    mp.spawn(_cae_worker, args=(2, _free_port(), two, ckpt, shape, patch), nprocs=2, join=True)
    files = sorted(os.listdir(os.path.join(one, '0', '0')))
    assert files == sorted(os.listdir(os.path.join(two, '0', '0'))) and len(files) == 12 + 1
    for f in files:
        assert open(os.path.join(one, '0', '0', f), 'rb').read() == open(os.path.join(two, '0', '0', f), 'rb').read(), f
    assert np.array_equal(zarrio.decompress_image(one), zarrio.decompress_image(two))


def _grad_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cnn_autoencoder_amd import train
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)  # identical replicas
    net = torch.nn.Sequential(torch.nn.Linear(7, 300), torch.nn.Linear(300, 5))
    for p in net.parameters():
        p.grad = torch.full_like(p, float(rank + 1))
    net[1].bias.grad = None  # a parameter that received no gradient on this rank
    reducer = train.GradReducer(net.parameters(), bucket_bytes=4096)  # several buckets
    reducer.reduce()
    torch.save([p.grad.clone() for p in net.parameters()], os.path.join(out_dir, f'g{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def _overlap_worker(rank, world, port, out_dir):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from cnn_autoencoder_amd import train
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 64), torch.nn.Tanh(), torch.nn.Linear(64, 64), torch.nn.Tanh(),
                              torch.nn.Linear(64, 3))
    ref = [p.detach().clone() for p in net.parameters()]
    reducer = train.GradReducer(list(net.parameters()), bucket_bytes=8192)  # several buckets
    gen = torch.Generator().manual_seed(11)
    out = []
    for step in range(2):
        x = torch.randn(8, 40, generator=gen)  # the global batch; this rank takes its half
        lo = 4 * rank
        net(x[lo:lo + 4]).pow(2).sum().backward()  # hooks fire here: buckets start their all-reduce during the backward
        started = reducer.launched_in_backward
        reducer.reduce()
        full = torch.nn.Sequential(*[type(m)(m.in_features, m.out_features) if isinstance(m, torch.nn.Linear) else torch.nn.Tanh()
                                     for m in net])
        for q, r in zip(full.parameters(), ref):
            q.data.copy_(r)
        full(x).pow(2).sum().backward()
        out.append(dict(started=started, got=[p.grad.clone() for p in net.parameters()],
                        want=[0.5 * q.grad for q in full.parameters()]))
        for p in net.parameters():
            p.grad = None
    torch.save(out, os.path.join(out_dir, f'o{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_starts_during_the_backward(tmp_path):
    """GradReducer's hooks launch a bucket's all-reduce as soon as its last gradient exists (overlap with the rest of the
    backward); the averaged gradients equal half the full-batch gradient; buckets persist over steps."""
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(str(tmp_path / f'o{r}.pt'), weights_only=False) for r in range(world)]
    for step in range(2):
        assert outs[0][step]['started'] >= 2 * (step + 1)  # >= 2 of the buckets per step went out from a hook
        for a, b, w in zip(outs[0][step]['got'], outs[1][step]['got'], outs[0][step]['want']):
            assert torch.equal(a, b)
            assert torch.allclose(a, w, rtol=1e-5, atol=1e-6)


def test_two_ranks_average_their_gradients(tmp_path):
    """train.GradReducer (data-parallel gradient all-reduce replacing nn.DataParallel, _autoencoders.py:517): every
    rank ends with the mean gradient, over several buckets, missing gradients counted as zero."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    g = [torch.load(str(tmp_path / f'g{r}.pt'), weights_only=False) for r in range(world)]
    for a, b in zip(*g):
        assert torch.equal(a, b)
    assert torch.allclose(g[0][0], torch.full_like(g[0][0], 1.5))
    assert torch.allclose(g[0][3], torch.zeros_like(g[0][3]))


def _train_worker(rank, world, port, out_dir, steps):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import cnn_autoencoder_amd as cae
    from cnn_autoencoder_amd import criteria, synth, train
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    cfg = dict(synth.CANONICAL, channels_net=32, channels_bn=48, compression_level=3)
    model = cae.autoencoder_from_state_dict(synth.synthetic_state(cfg, seed=31), train=True)
    opts = train.setup_optim(model, learning_rate=1e-3, aux_learning_rate=1e-2)
    reducer = train.GradReducer([p for k in ('encoder', 'decoder', 'fact_ent') for p in model[k].parameters()])
    criterion = criteria.GeneralLoss(distortion_lambda=0.01)
    gen = torch.Generator().manual_seed(5)
    for _ in range(steps):
        x = torch.rand(4, 3, 32, 48, generator=gen)       # the global batch; this rank trains on its half
        noise = torch.rand(4, 48, 4, 6, generator=gen) - 0.5
        lo, hi = (0, 4) if world == 1 else (2 * rank, 2 * rank + 2)
        model['fact_ent'].module.fixed_noise = noise[lo:hi]
        train.train_step(x[lo:hi].cuda(), model, criterion, opts, reducer=reducer)
    state = {k: {n: p.detach().cpu() for n, p in model[k].named_parameters()} for k in model}
    torch.save(state, os.path.join(out_dir, f'w{world}r{rank}.pt'))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_training_equals_the_full_batch(tmp_path):
    """Data-parallel training (config 5): two ranks, each on half of the batch, gradients averaged by GradReducer
    (gloo here, RCCL on a multi-GPU node) == one process on the whole batch; replicas stay identical."""
    steps = 3
    mp.spawn(_train_worker, args=(1, _free_port(), str(tmp_path), steps), nprocs=1, join=True)
    mp.spawn(_train_worker, args=(2, _free_port(), str(tmp_path), steps), nprocs=2, join=True)
    one = torch.load(str(tmp_path / 'w1r0.pt'), weights_only=False)
    two = [torch.load(str(tmp_path / f'w2r{r}.pt'), weights_only=False) for r in range(2)]
    for k in one:
        for n in one[k]:
            assert torch.equal(two[0][k][n], two[1][k][n]), f'replicas diverged: {k}.{n}'
            a, b = one[k][n], two[0][k][n]
            assert float((a - b).abs().max()) <= 2e-3 * max(float(a.abs().max()), 1e-3), f'{k}.{n}'
