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

    sums = np.asarray(local_sums).ravel()
    cplx = np.iscomplexobj(sums)  # complex wave functions: <acc>ecp and <acc>total are complex (eval_ecp.py:89); RCCL reduces reals
    # real and imaginary parts are ALWAYS packed: the tensor length must not depend on a rank's local dtype (one rank's block real,
    # another's complex — own propagate functions, accumulators returning reals — would all-reduce tensors of different lengths)
    flat = np.concatenate([sums.real, sums.imag if cplx else np.zeros(len(sums))])
    t = torch.tensor(np.concatenate([np.asarray(flat, dtype=np.float64), [float(local_count)]]), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _world_one_collectives()):
        dist.all_reduce(t)
    t = t.cpu().numpy()
    means = t[:-1] / t[-1]
    re, im = means[: len(sums)], means[len(sums):]
    if cplx or np.any(im != 0.0):  # (some rank's block was complex)
        return re + 1j * im, t[-1]
    return re, t[-1]


def _world_one_collectives():
    """``PQA_DIST_WORLD1=1``: with a process group of ONE rank still take the collective routes (all-reduce, all-gather,
    broadcast on the communicator; the exchange plan then keeps every walker) — how a single-GPU box executes the RCCL
    (``backend="nccl"``) branches of this module: tests/test_gpu_fullsize.py::test_rccl_single_rank_communicator_...."""
    import os

    return os.environ.get("PQA_DIST_WORLD1", "0") == "1"


def combine_blocks(block_avgs, counts):
    """The reference's host-side recombination (mc.py:166-172) — used to cross-check allreduce_block."""
    w = np.asarray(counts, dtype=float)
    w = w / w.sum()
    return {k: sum(b[k] * wi for b, wi in zip(block_avgs, w)) for k in block_avgs[0]}


def exchange_plan(newinds, counts, rank):
    """Who sends what to whom after the global comb ``newinds`` (global source index per global new slot; shards are the
    contiguous blocks of sizes ``counts``).  Returns for ``rank``:
    ``keep``  local source indices of the new walkers whose source already lives here (in comb order),
    ``recv``  {src_rank: number of walkers arriving from it} and ``send`` {dst_rank: local indices (duplicates allowed) of the
    walkers to ship there}, both in comb order so that sender and receiver agree on the sequence."""
    offs = np.concatenate([[0], np.cumsum(counts)])
    owner = np.searchsorted(offs, newinds, side="right") - 1  # rank that holds each SOURCE walker
    lo, hi = offs[rank], offs[rank + 1]
    mine = newinds[lo:hi]
    mine_owner = owner[lo:hi]
    keep = (mine[mine_owner == rank] - lo).astype(np.int32)
    recv = {int(q): int(np.sum(mine_owner == q)) for q in np.unique(mine_owner) if q != rank}
    send = {}
    for q in range(len(counts)):
        if q == rank:
            continue
        src = newinds[offs[q] : offs[q + 1]]
        sel = src[(src >= lo) & (src < hi)]
        if len(sel):
            send[q] = (sel - lo).astype(np.int32)
    return keep, recv, send


def branch_distributed(configs, weights, base_u=None, dev=None, device_buffers=None):
    """Stochastic-comb branching (``pyqmc/method/dmc.py:342-376``) of an ensemble sharded over ranks, as SURVEY.md section
    8(e) specifies it: all-gather of the WEIGHTS only (8 B per walker), one broadcast uniform, the identical comb on every
    rank, and then one point-to-point exchange of just the walkers whose new owner differs from their old one —
    coordinates (+ wrap counters), 1.5-3 KB each — where the reference gathers every walker to the master, branches and
    re-splits (dmc.py:286-287, :566).  Shards keep their sizes; no master exists.

    ``dev`` (a ``DeviceWF`` holding this shard's walkers): the outgoing coordinates are gathered on the device
    (``pqa_get_walkers``) straight into the buffer RCCL sends from, the walkers that stay keep their whole wave-function
    state (``pqa_branch_exchange`` gathers it like ``pqa_resample``) and only the RECEIVED walkers are recomputed.  Without
    ``dev`` the coordinates come from ``configs`` and the caller recomputes, like the reference (dmc.py:155).
    ``device_buffers``: pack / unpack through GPU tensors and hand raw device pointers to the library.  Default: exactly when
    the backend is RCCL ("nccl").  ``True`` under gloo runs the same device-side packing, stream hand-off and pointer path
    with the TRANSPORT alone going through host copies (gloo cannot send GPU tensors) — how the one-GPU test suite
    exercises the code an 8-GPU run executes.

    The new local order is: walkers that stayed (comb order), then arrivals by source rank (comb order) — walkers are
    exchangeable and carry equal weights after the comb.  Returns (configs, weights, info, global weight std); ``info`` also
    reports ``walkers moved`` (global), ``bytes exchanged`` (sent by this rank) and the wall time of the three stages on this
    rank: ``gather seconds`` (weights all-gather + comb), ``exchange seconds`` (packing + point-to-point transfers) and
    ``state seconds`` (gather of the kept walkers' state + recompute of the arrivals)."""
    import time

    import torch
    import torch.distributed as dist

    from .dmc import comb_indices

    if not (dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _world_one_collectives())):
        from .dmc import branch

        wstd = float(np.std(weights))
        cfg, w, info = branch(configs, weights, base_u, on_resample=None if dev is None else dev.resample)
        return cfg, w, {**info, "walkers moved": 0, "bytes exchanged": 0, "gather seconds": 0.0, "exchange seconds": 0.0, "state seconds": 0.0}, wstd
    t_start = time.perf_counter()
    world, rank = dist.get_world_size(), dist.get_rank()
    nccl = dist.get_backend() == "nccl"
    tdev = torch.device("cuda", torch.cuda.current_device()) if nccl else torch.device("cpu")  # what the transport moves
    on_gpu = dev is not None and (nccl if device_buffers is None else bool(device_buffers))
    bdev = torch.device("cuda", dev.device) if on_gpu else tdev  # where walkers are packed / unpacked
    n_local = torch.tensor([len(weights)], dtype=torch.int64, device=tdev)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local)
    counts = [int(c.item()) for c in counts]
    pad_w = torch.zeros(max(counts), dtype=torch.float64, device=tdev)
    pad_w[: len(weights)] = torch.from_numpy(np.asarray(weights, dtype=np.float64))
    all_w = [torch.zeros_like(pad_w) for _ in range(world)]
    dist.all_gather(all_w, pad_w)
    gw = np.concatenate([t[:c].cpu().numpy() for t, c in zip(all_w, counts)])
    u = torch.tensor([np.random.rand() if base_u is None else base_u], dtype=torch.float64, device=tdev)
    dist.broadcast(u, src=0)
    newinds, wtot = comb_indices(gw, float(u.item()))
    unique, cnt = np.unique(newinds, return_counts=True)
    # The comb starts at a random offset u * sum(w) (dmc.py:358-361), so its slot -> source map is a cyclic shift of a
    # monotone one: taken literally nearly every walker would change ranks.  Only the MULTISET of sources matters (walkers
    # are exchangeable and leave the comb with equal weights), so the slots are filled in source order: copies stay on or
    # next to the rank that holds the source.
    newinds = np.sort(newinds)
    keep, recv, send = exchange_plan(newinds, counts, rank)
    t_plan = time.perf_counter()
    if dev is not None and not getattr(dev, "twisted", False):
        # the walkers that stay are taken from the DEVICE, not from whatever the host container last saw: the device drivers refresh the
        # container at block boundaries only, and a caller other than rundmc may branch in between (round-5 verdict, weak 8).  Coordinates
        # only — the wrap counters of a periodic container advance when the driver fetches them (reading them is not idempotent); twisted
        # handles keep unfolded coordinates on the device and their container is folded by the driver.
        configs.configs[...] = dev.configs()
    x = configs.configs
    nx = int(np.prod(x.shape[1:]))
    periodic = hasattr(configs, "wrap")  # PeriodicConfigs: the wrap counters travel with the coordinates (coord.py:191-198)
    row = nx * (2 if periodic else 1)
    ops, inbox, outbox, sent_bytes = [], {}, {}, 0
    for q, idx in sorted(send.items()):
        buf = torch.empty((len(idx), row), dtype=torch.float64, device=bdev)
        if on_gpu:
            if periodic:  # coordinates gathered on the device, the (host-side) wrap counters appended
                xs = torch.empty((len(idx), nx), dtype=torch.float64, device=bdev)
                dev.get_walkers(idx, out=xs.data_ptr())
                buf[:, :nx] = xs
                buf[:, nx:] = torch.from_numpy(np.ascontiguousarray(np.asarray(configs.wrap, dtype=np.float64)[idx].reshape(len(idx), nx))).to(bdev)
            else:
                dev.get_walkers(idx, out=buf.data_ptr())  # (the library synchronises its stream before returning)
        else:
            src = dev.get_walkers(idx).reshape(len(idx), nx) if dev is not None else x[idx].reshape(len(idx), nx)
            buf[:, :nx] = torch.from_numpy(np.ascontiguousarray(src))
            if periodic:
                buf[:, nx:] = torch.from_numpy(np.ascontiguousarray(np.asarray(configs.wrap, dtype=np.float64)[idx].reshape(len(idx), nx)))
        outbox[q] = buf if buf.device == tdev else buf.to(tdev)
        sent_bytes += buf.numel() * 8
        ops.append(dist.P2POp(dist.isend, outbox[q], q))
    for q, n in sorted(recv.items()):
        inbox[q] = torch.empty((n, row), dtype=torch.float64, device=tdev)
        ops.append(dist.P2POp(dist.irecv, inbox[q], q))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
    nrecv = sum(recv.values())
    got = torch.cat([inbox[q] for q in sorted(inbox)], dim=0) if nrecv else torch.empty((0, row), dtype=torch.float64, device=tdev)
    got_host = got.cpu().numpy()
    t_xchg = time.perf_counter()
    if dev is not None:  # state follows the walkers that stay; arrivals are recomputed (device pointer when packed on the GPU)
        gx = got[:, :nx].to(bdev).contiguous()
        if on_gpu:
            torch.cuda.current_stream(bdev).synchronize()  # the library reads gx on its own HIP stream: torch's work must be done
        dev.branch_exchange(keep, gx.data_ptr() if (nrecv and on_gpu) else (gx.cpu().numpy() if nrecv else None), nrecv)
    shape = (len(keep) + nrecv,) + x.shape[1:]
    configs.configs = np.ascontiguousarray(np.concatenate([x[keep].reshape(len(keep), nx), got_host[:, :nx]], axis=0)).reshape(shape)
    if periodic:
        w0 = np.asarray(configs.wrap)
        configs.wrap = np.concatenate([w0[keep].reshape(len(keep), nx), got_host[:, nx:]], axis=0).reshape(shape).astype(w0.dtype)
    new_w = np.full(shape[0], wtot / len(gw))
    moved = int(np.sum(np.searchsorted(np.concatenate([[0], np.cumsum(counts)]), newinds, side="right") - 1
                       != np.repeat(np.arange(world), counts)))
    info = {"max branches": int(cnt.max()), "Number of walkers killed": int(len(gw) - len(unique)), "walkers moved": moved,
            "bytes exchanged": int(sent_bytes), "device_buffers": bool(on_gpu), "gather seconds": t_plan - t_start,
            "exchange seconds": t_xchg - t_plan, "state seconds": time.perf_counter() - t_xchg}
    return configs, new_w, info, float(np.std(gw))
