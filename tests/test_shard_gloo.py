"""The N > 1 path on CPU: world_size-2 gloo processes exercising the batch sharder
(partition, optional gather, scalar combine, max-over-ranks timing)."""
import os
import socket
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from conftest import ROOT


def test_shard_bounds_tile_the_range():
    from nitorch_fastmath_amd.shard import shard_bounds
    for n in (0, 1, 7, 8, 100_000_001):
        for world in (1, 2, 3, 8):
            b = [shard_bounds(n, r, world) for r in range(world)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, q):
    import sys
    sys.path.insert(0, ROOT)
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from nitorch_fastmath_amd import shard
        import oracle as O
        # every rank builds the same global problem, works on its own shard only
        rng = np.random.default_rng(0)
        M = 4
        G = rng.standard_normal((n, M, M))
        A = G @ G.transpose(0, 2, 1) / M + np.eye(M)
        iu = [(i, j) for i in range(M) for j in range(i + 1, M)]
        mat = np.concatenate([np.stack([A[:, i, i] for i in range(M)], -1),
                              np.stack([A[:, i, j] for i, j in iu], -1)], -1).astype(np.float32)
        vec = rng.standard_normal((n, M)).astype(np.float32)
        full = O.sym_solve(mat, vec)
        lo, hi = shard.shard_bounds(n, rank, world)
        mine = torch.from_numpy(O.sym_solve(mat[lo:hi], vec[lo:hi]))   # stand-in for the GPU kernel
        assert tuple(shard.shard_of(torch.from_numpy(vec), rank, world).shape) == (hi - lo, M)
        got = shard.gather_outputs(mine, n)
        ok_gather = bool(np.array_equal(got.numpy(), full))
        # sharded full reduction + scalar combine
        x = rng.standard_normal(10 * n).astype(np.float32)
        x[::97] = np.nan
        l2, h2 = shard.shard_bounds(x.size, rank, world)
        part = torch.tensor(float(O.reduce('nansum', x[l2:h2], out_f64=True)), dtype=torch.float64)
        tot = shard.combine_scalar(part, 'nansum')
        ok_sum = abs(float(tot) - float(O.reduce('nansum', x, out_f64=True))) < 1e-9
        pmax = torch.tensor(float(O.reduce('nanmax', x[l2:h2])), dtype=torch.float64)
        ok_max = float(shard.combine_scalar(pmax, 'nanmax')) == float(O.reduce('nanmax', x))
        t = shard.max_over_ranks(1.0 + rank)
        q.put((rank, ok_gather, ok_sum, ok_max, t))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('n', [1001, 64])
def test_world2_gloo(oracle, n):
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_gather, ok_sum, ok_max, t in res:
        assert ok_gather and ok_sum and ok_max, (rank, ok_gather, ok_sum, ok_max)
        assert t == float(world)       # slowest rank = 1.0 + (world - 1)
