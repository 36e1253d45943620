"""Walker sharding and block reductions for multi-GPU runs (one process per GPU).

The reference parallelises by splitting the walkers over workers (``configs.split``,
``pyqmc/configurations/coord.py:72-80``; ``vmc_parallel`` ``pyqmc/method/mc.py:156-173``) and
re-weighting the per-worker block averages by walker count (:166-172).  Here each rank owns a
contiguous shard for the whole run and the only exchange of the VMC path is one all-reduce of
``[sum_w q_k ..., count]`` per block — RCCL over xGMI on GPUs (``backend="nccl"``), gloo in the
CPU tests.  There is no data-path collective: walkers never move between ranks in VMC.
"""

import numpy as np


def shard_bounds(nconfig, world_size):
    """Contiguous shard [lo, hi) per rank with ``np.array_split`` semantics (remainder to low ranks)."""
    base, rem = divmod(int(nconfig), int(world_size))
    sizes = [base + (r < rem) for r in range(world_size)]
    hi = np.cumsum(sizes)
    return [(int(h - s), int(h)) for s, h in zip(sizes, hi)]


def shard(configs_array, rank, world_size):
    lo, hi = shard_bounds(len(configs_array), world_size)[rank]
    return configs_array[lo:hi]


def allreduce_block(local_sums, local_count, device=None):
    """Global walker-weighted means of per-rank sums.

    local_sums: (k,) sums over this rank's walkers (and steps); local_count: number of samples behind
    them.  Returns (means (k,), total_count).  Works without an initialised process group (single rank).
    ``device``: where the reduction tensor lives; default: the current GPU under the RCCL backend ("nccl" only reduces
    device tensors), the host otherwise.
    """
    import torch
    import torch.distributed as dist

    if device is None and dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())

    t = torch.tensor(np.concatenate([np.asarray(local_sums, dtype=np.float64).ravel(), [float(local_count)]]),
                     dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    t = t.cpu().numpy()
    return t[:-1] / t[-1], t[-1]


def combine_blocks(block_avgs, counts):
    """The reference's host-side recombination (mc.py:166-172) — used to cross-check allreduce_block."""
    w = np.asarray(counts, dtype=float)
    w = w / w.sum()
    return {k: sum(b[k] * wi for b, wi in zip(block_avgs, w)) for k in block_avgs[0]}


def branch_distributed(configs, weights, base_u=None):
    """Stochastic-comb branching (``pyqmc/method/dmc.py:342-376``) of an ensemble sharded over ranks.

    The reference gathers all walkers to the master, branches and re-splits (dmc.py:286-287, :566).  Here every
    rank all-gathers the weights (8 B per walker) and the coordinates, rank 0 broadcasts the single uniform of
    the comb, every rank computes the same global resampling indices and keeps the slice that belongs to its
    shard — so shards stay exactly the size they had and no master exists.  Wave-function internals are not
    exchanged: like the reference (dmc.py:155) the caller recomputes them at the next propagate.
    Returns (configs, weights, info, global weight std).
    """
    import torch
    import torch.distributed as dist

    from .dmc import comb_indices

    x = np.ascontiguousarray(configs.configs)
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        from .dmc import branch

        wstd = float(np.std(weights))
        return (*branch(configs, weights, base_u), wstd)
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    n_local = torch.tensor([len(weights)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    nx = int(np.prod(x.shape[1:]))
    payload = x.reshape(len(weights), nx)
    periodic = hasattr(configs, "wrap")  # PeriodicConfigs: the wrap counters travel with the coordinates (coord.py:191-198)
    if periodic:
        payload = np.concatenate([payload, np.asarray(configs.wrap, dtype=np.float64).reshape(len(weights), nx)], axis=1)
    nmax, row = max(counts), payload.shape[1]
    pad_w = torch.zeros(nmax, dtype=torch.float64, device=dev)
    pad_w[: len(weights)] = torch.from_numpy(np.asarray(weights, dtype=np.float64))
    pad_x = torch.zeros(nmax, row, dtype=torch.float64, device=dev)
    pad_x[: len(weights)] = torch.from_numpy(np.ascontiguousarray(payload))
    all_w = [torch.zeros_like(pad_w) for _ in range(world)]
    all_x = [torch.zeros_like(pad_x) for _ in range(world)]
    dist.all_gather(all_w, pad_w)
    dist.all_gather(all_x, pad_x)
    gw = np.concatenate([t[:c].cpu().numpy() for t, c in zip(all_w, counts)])
    gx = np.concatenate([t[:c].cpu().numpy() for t, c in zip(all_x, counts)])
    u = torch.tensor([np.random.rand() if base_u is None else base_u], dtype=torch.float64, device=dev)
    dist.broadcast(u, src=0)
    newinds, wtot = comb_indices(gw, float(u.item()))
    unique, cnt = np.unique(newinds, return_counts=True)
    lo = int(np.sum(counts[:rank]))
    mine = newinds[lo : lo + counts[rank]]
    configs.configs = np.ascontiguousarray(gx[mine][:, :nx]).reshape((len(mine),) + x.shape[1:])
    if periodic:
        configs.wrap = np.ascontiguousarray(gx[mine][:, nx:]).reshape((len(mine),) + x.shape[1:])
    new_w = np.full(len(mine), wtot / len(gw))
    info = {"max branches": int(cnt.max()), "Number of walkers killed": int(len(gw) - len(unique))}
    return configs, new_w, info, float(np.std(gw))
