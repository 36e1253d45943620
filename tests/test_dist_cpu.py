"""Multi-rank path on CPU: world_size=2, gloo.  The ranks shard an ensemble, compute block sums
with the ORACLE standing in for the per-rank engine (no GPU here), and all-reduce them; the result
must equal the single-process answer and the reference's own recombination rule (mc.py:166-172)."""

import os
import socket
import sys

import numpy as np
import pytest

import helpers
from pyqmc_amd import dist as pdist
from pyqmc_amd import systems
from pyqmc_amd.configs import OpenConfigs


def test_shard_bounds_match_array_split():
    for n in (1, 7, 100, 4097):
        for ws in (1, 2, 3, 8):
            parts = np.array_split(np.arange(n), ws)
            b = pdist.shard_bounds(n, ws)
            assert [(int(p[0]) if len(p) else lo, len(p)) for p, (lo, hi) in zip(parts, b)] == [(lo, hi - lo) for lo, hi in b]
            assert b[0][0] == 0 and b[-1][1] == n


def _worker(rank, world, port, start, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import energy as oenergy

        mol = systems.water()
        wf = helpers.oracle_wf(mol, systems.random_mf(mol))
        mine = OpenConfigs(pdist.shard(start, rank, world).copy())
        wf.recompute(mine)
        ke, g2 = oenergy.kinetic(mine, wf)
        ee, ei, ii = oenergy.coulomb(mol, mine)
        sums = np.array([ke.sum(), ee.sum(), ei.sum(), g2.sum()])
        means, count = pdist.allreduce_block(sums, len(ke))
        q.put((rank, means, count, sums, len(ke)))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_block_reduction():
    import torch.multiprocessing as mp

    mol = systems.water()
    start = systems.initial_guess(mol, 9, rng=np.random.default_rng(2)).configs  # 9 walkers -> shards of 5 and 4
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, start, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process answer
    from oracle import energy as oenergy

    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    cfg = OpenConfigs(start.copy())
    wf.recompute(cfg)
    ke, g2 = oenergy.kinetic(cfg, wf)
    ee, ei, _ = oenergy.coulomb(mol, cfg)
    ref = np.array([ke.mean(), ee.mean(), ei.mean(), g2.mean()])
    for rank, means, count, sums, n in res:
        assert count == 9 and np.allclose(means, ref, rtol=1e-13)
    # the reference's recombination of per-worker block averages gives the same numbers
    blocks = [dict(zip("abcd", r[3] / r[4])) for r in res]
    comb = pdist.combine_blocks(blocks, [r[4] for r in res])
    assert np.allclose([comb[k] for k in "abcd"], ref, rtol=1e-13)
    assert [r[4] for r in res] == [5, 4]
