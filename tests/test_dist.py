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
