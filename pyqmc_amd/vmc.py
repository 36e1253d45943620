"""VMC drivers.

``vmc_worker`` has the signature and return contract of the reference's
``pyqmc.method.mc.vmc_worker`` (``mc.py:102-153``): ``(block_avg dict, configs)`` with keys
``<acc><quantity>``, ``acceptance``, ``"move time"``, ``"accumulator time"``.

Two ways to run it:
  * fused (default when the wave function lives on one device handle and the only accumulators
    are ``EnergyAccumulator``s): the whole electron loop and the energy evaluation run on the
    GPU (``pqa_vmc_sweeps``); random numbers come from the device Philox stream seeded from
    ``numpy.random`` (or from explicit tapes for trajectory-level parity tests);
  * protocol: the reference's Python loop, calling the wave-function protocol once per electron
    — the drop-in path an unmodified driver takes.  Host work here is only control flow and the
    random numbers, exactly as in the reference.
"""

import time

import numpy as np

from .energy import KEYS, EnergyAccumulator


def limdrift(g, cutoff=1):
    """mc.py:76-89."""
    tot = np.linalg.norm(g, axis=1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return np.where((tot > cutoff)[:, None], cutoff * g / tot[:, None], g)


def _fusable(wf, accumulators):
    dev = wf.fused_device() if hasattr(wf, "fused_device") else None
    return dev if dev is not None and all(isinstance(a, EnergyAccumulator) for a in accumulators.values()) else None


def vmc_worker(wf, configs, tstep, nsteps, accumulators, fused=None, tapes=None, seed=None, state_current=False, fetch_configs=True):
    """One block of ``nsteps`` sweeps (mc.py:102-153): returns (block averages, configs).

    ``state_current`` / ``fetch_configs`` are for block loops on the fused device path (``vmc`` below): the reference's worker
    is a function of (wf, configs) and so recomputes the wave function from ``configs`` on entry and hands the walkers back
    on exit — 100 MB each way over PCIe plus a full recompute per block at 65536 walkers of the 64-electron system.  A caller
    that knows the device already holds the state of ``configs`` (it ran the previous block and touched nothing since) passes
    ``state_current=True``; one that does not need the walkers on the host after this block passes ``fetch_configs=False``
    (open systems only: periodic containers also carry the block's wrap counters)."""
    dev = _fusable(wf, accumulators) if fused in (None, True) else None
    if fused is True and dev is None:
        raise TypeError("fused VMC needs a pyqmc_amd wave function on one device handle and EnergyAccumulator only")
    if dev is None:
        return _vmc_worker_protocol(wf, configs, tstep, nsteps, accumulators)
    tapes = tapes or {}
    if seed is None:
        seed = int(np.random.randint(0, 2**31 - 1))
    if not state_current:
        wf.recompute(configs)
    block_avg = {}
    thr = next(iter(accumulators.values())).threshold if accumulators else 10.0
    if dev.pbc and accumulators:
        dev.set_ewald(**next(iter(accumulators.values()))._ewald_kws)
    t0 = time.perf_counter()
    acc, en, rec = dev.vmc_sweeps(tstep, nsteps, gauss=tapes.get("gauss"), unif=tapes.get("unif"), threshold=thr,
                                  ecp_rot=tapes.get("ecp_rot"), ecp_unif=tapes.get("ecp_unif"), seed=seed,
                                  energy=bool(accumulators), record="record" in tapes)
    t1 = time.perf_counter()
    if "record" in tapes:
        tapes["record"].append(rec)
    for name in accumulators:
        for i, k in enumerate(KEYS):
            block_avg[name + k] = np.sum(en[:, i]) / nsteps
    block_avg["acceptance"] = acc[-1]
    # the fused kernel interleaves moves and energy; report the split the reference reports as one number each
    block_avg["move time"] = (t1 - t0) / nsteps
    block_avg["accumulator time"] = 0.0
    if not fetch_configs and not dev.pbc:
        return block_avg, configs  # (stale on the host until a later block fetches them)
    if getattr(dev, "twisted", False):  # the handle keeps true (unfolded) coordinates: fold them back into the container
        from .configs import enforce_pbc

        configs.configs[...], configs.wrap[...] = enforce_pbc(configs.lvecs, dev.configs())
    else:
        configs.configs[...] = dev.configs()
        if dev.pbc:  # walkers stay folded into the cell; their wrap counters advance (coord.py:180-189)
            configs.wrap += dev.wrap_delta()
    return block_avg, configs


def _vmc_worker_protocol(wf, configs, tstep, nsteps, accumulators):
    """Line-for-line control flow of mc.py:102-153 over the protocol."""
    nconf, nelec, _ = configs.configs.shape
    block_avg = {}
    wf.recompute(configs)
    for _ in range(nsteps):
        acc = 0.0
        t0 = time.perf_counter()
        for e in range(nelec):
            g, _, _ = wf.gradient_value(e, configs.electron(e))
            grad = limdrift(np.real(g.T))
            gauss = np.random.normal(scale=np.sqrt(tstep), size=(nconf, 3))
            newcoorde = configs.make_irreducible(e, configs.configs[:, e, :] + gauss + grad * tstep)
            g, new_val, saved = wf.gradient_value(e, newcoorde)
            new_grad = limdrift(np.real(g.T))
            forward = np.sum(gauss**2, axis=1)
            backward = np.sum((gauss + tstep * (grad + new_grad)) ** 2, axis=1)
            t_prob = np.exp(1 / (2 * tstep) * (forward - backward))
            ratio = np.abs(new_val) ** 2 * t_prob
            accept = ratio > np.random.rand(nconf)
            configs.move(e, newcoorde, accept)
            wf.updateinternals(e, newcoorde, configs, mask=accept, saved_values=saved)
            acc += np.mean(accept) / nelec
        t1 = time.perf_counter()
        for k, accumulator in accumulators.items():
            dat = accumulator.avg(configs, wf)
            for m, res in dat.items():
                block_avg[k + m] = block_avg.get(k + m, 0.0) + res / nsteps
        t2 = time.perf_counter()
        block_avg["acceptance"] = acc
        block_avg["move time"] = t1 - t0
        block_avg["accumulator time"] = t2 - t1
    return block_avg, configs


def vmc(wf, configs, nblocks=10, nsteps_per_block=10, tstep=0.5, accumulators=None, verbose=False, seed=None, fused=None,
        hdf_file=None, continue_from=None, recompute_every=10):
    """Block loop of ``pyqmc.method.mc.vmc`` (mc.py:176-274): returns (dict of arrays over the blocks run, configs).
    ``hdf_file``: block output in the reference's on-disk layout (``pyqmc_amd.blockfile``: HDF5 when h5py exists, NumPy
    archives otherwise); an existing file — or ``continue_from`` — restarts from its walkers at ``block[-1] + 1``, and
    ``nblocks`` counts the blocks of all calls together, as in the reference (mc.py:223-243)."""
    from .blockfile import BlockFile

    accumulators = accumulators or {}
    out = None if hdf_file is None else BlockFile(hdf_file)
    source = out if continue_from is None else BlockFile(continue_from)
    if continue_from is not None:
        if not source.exists():
            raise RuntimeError(f"cannot continue from {continue_from}; the file does not exist!")
        if out is not None and out.exists():
            raise RuntimeError(f"continue_from is not None but hdf_file={hdf_file} already exists! Delete or rename {hdf_file} and try again.")
    first = 0
    if source is not None and source.exists() and source.last_block() is not None:
        first = source.last_block() + 1
        source.load_walkers(configs)
    df = {}
    dev = _fusable(wf, accumulators) if fused in (None, True) else None
    current = False  # the device holds the wave-function state of the walkers it moved in the previous block
    for block in range(first, nblocks):
        # fused path: the walkers stay on the device from block to block; the state is rebuilt from the (fetched) walkers every
        # `recompute_every` blocks to bound the Sherman-Morrison round-off, and the host copy is refreshed when a block is
        # written to disk, before such a rebuild, and at the end
        rebuild_next = dev is not None and (block + 1 - first) % recompute_every == 0
        fetch = out is not None or block == nblocks - 1 or rebuild_next
        blk, configs = vmc_worker(wf, configs, tstep, nsteps_per_block, accumulators, fused=fused,
                                  seed=None if seed is None else seed + block, state_current=current, fetch_configs=fetch)
        current = dev is not None and not rebuild_next
        blk["block"] = block
        blk["nconfig"] = nsteps_per_block * configs.configs.shape[0]
        if out is not None:
            out.append(blk, {"tstep": tstep}, configs)
        if verbose:
            print(f"block {block}: " + ", ".join(f"{k}={np.real(v):.6g}" for k, v in blk.items() if "total" in k or k == "acceptance"))
        for k, v in blk.items():
            df.setdefault(k, []).append(v)
    return {k: np.asarray(v) for k, v in df.items()}, configs
