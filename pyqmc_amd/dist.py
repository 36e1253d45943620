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
    """
    import torch
    import torch.distributed as dist

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
