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


def _c5_like_weights(W, seed):
    """Weights as a C5 block leaves them (profiles/r02_branch_traffic.jsonl: sigma(w)/mean(w) = 0.18-0.19 after 5 steps at
    tstep 0.02): log-normal around 1 with that spread."""
    rng = np.random.default_rng(seed)
    return np.exp(0.185 * rng.standard_normal(W) - 0.5 * 0.185**2)


def _ws8_worker(rank, world, port, W, q):
    import torch.distributed as dist

    from pyqmc_amd.configs import OpenConfigs

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = pdist.shard_bounds(W, world)[rank]
        x = np.arange(W * 6, dtype=float).reshape(W, 2, 3)[lo:hi]  # walker w carries its global index in every coordinate block
        w = _c5_like_weights(W, 5)[lo:hi]
        cfg, neww, info, wstd = pdist.branch_distributed(OpenConfigs(x.copy()), w.copy(), base_u=0.6180339887)
        q.put((rank, cfg.configs, neww, info))
    finally:
        dist.destroy_process_group()


def test_distributed_branch_eight_ranks_c5_like_weights():
    """The first 8-rank run of the branching step should be boring: world size 8 (gloo), 32768 walkers with weights spread like
    a C5 block's.  Every rank keeps its shard size, the ranks together hold exactly the single-process comb's survivors, and
    the walkers that crossed ranks are the population drift only — 0.1-1 % of the ensemble (DESIGN.md section 7 measured
    55-149 of 32768 on real C5 weights), each counted once, with nothing else on the wire."""
    import torch.multiprocessing as mp

    from pyqmc_amd import dmc

    W, world = 32768, 8
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ws8_worker, args=(r, world, port, W, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    gw = _c5_like_weights(W, 5)
    newinds = np.sort(dmc.comb_indices(gw, 0.6180339887)[0])
    got = np.sort(np.concatenate([r[1][:, 0, 0] / 6 for r in res]).astype(np.int64))  # global source index of every survivor
    assert np.array_equal(got, newinds)
    bounds = pdist.shard_bounds(W, world)
    moved = res[0][3]["walkers moved"]
    assert all(r[3]["walkers moved"] == moved for r in res)
    assert 0.001 * W < moved < 0.01 * W, moved
    assert sum(r[3]["bytes exchanged"] for r in res) == moved * 6 * 8  # coordinates of the re-assigned walkers, nothing else
    for r, (lo, hi) in zip(res, bounds):
        assert len(r[1]) == hi - lo and np.allclose(r[2], gw.sum() / W, rtol=1e-14)
        # most of a rank's new walkers were already its own: arrivals are a small minority on every rank
        own = np.sum((r[1][:, 0, 0] / 6 >= lo) & (r[1][:, 0, 0] / 6 < hi))
        assert own > 0.97 * (hi - lo)


def _bench_logic_worker(rank, world, port, q):
    import argparse

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sys.path.insert(0, helpers.ROOT)
        import bench

        weak = bench.local_walkers(argparse.Namespace(scaling="weak", walkers=0), rank, world, 4096, 32768)
        weak_set = bench.local_walkers(argparse.Namespace(scaling="weak", walkers=1000), rank, world, 4096, 32768)
        strong = bench.local_walkers(argparse.Namespace(scaling="strong", walkers=0), rank, world, 4096, 32769)
        table = bench.rank_table(torch, dist, rank, 0, world)
        # the block reduction bench.py's timed region ends with: per-rank energy sums -> global walker-weighted means
        en = np.full((3, 6), 1.0 + rank)  # (steps, energy rows) of this rank's shard
        W = strong[0]
        means, count = pdist.allreduce_block(en.sum(axis=0) * W, en.shape[0] * W, device="cpu")
        tmax = torch.tensor([1.0 + rank], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        q.put((rank, weak, weak_set, strong, table, means, count, float(tmax.item())))
    finally:
        dist.destroy_process_group()


def test_bench_multi_rank_bookkeeping_two_ranks_gloo():
    """The parts of bench.py an N > 1 run adds to the N = 1 one — walkers per rank for weak / strong scaling (`local_walkers`),
    the per-rank device table and communicator size (`rank_table`), the block reduction and the max-over-ranks clock — driven
    at world size 2 over gloo (no GPU): what `SCALE_rNN.json` will be computed from is exercised before the first 8-GPU run."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_logic_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [(4096, 8192), (4096, 8192)]  # weak: the per-GPU default on every rank, total = per-GPU x ranks
    assert [r[2] for r in res] == [(1000, 2000), (1000, 2000)]
    assert [r[3] for r in res] == [(16385, 32769), (16384, 32769)]  # strong: the config's total split like configs.split (coord.py:72-80)
    for r in res:
        t = r[4]
        assert t["rccl_ranks"] == 2 and t["backend"] == "gloo" and [e["rank"] for e in t["ranks"]] == [0, 1]
        assert np.allclose(r[5], (1.0 * 16385 + 2.0 * 16384) / 32769) and r[6] == 3 * 32769 and r[7] == 2.0
