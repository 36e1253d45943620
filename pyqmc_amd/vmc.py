"""VMC drivers.

``vmc_worker`` has the signature and return contract of the reference's
``pyqmc.method.mc.vmc_worker`` (``mc.py:102-153``): ``(block_avg dict, configs)`` with keys
``<acc><quantity>``, ``acceptance``, ``"move time"``, ``"accumulator time"``.

The electron loop always runs on the GPU (``pqa_vmc_sweeps``): random numbers come from the device
Philox stream seeded from ``numpy.random`` (or from explicit tapes for trajectory-level parity
tests).  With ``EnergyAccumulator``s only, the energy evaluation is fused into the same call; any
other accumulator (density matrices, parameter gradients, ...) is called on the host after every
device sweep, on the walkers fetched from the device, and talks to the wave function through the
protocol entry points as it would in the reference.

The reference's own per-electron Python loop (``mc.py:115-137``) is NOT restated here: an unmodified
``pyqmc.method.mc.vmc_worker`` runs over these wave-function objects as they are,
and the parity tests drive the protocol entry points through ``tests/helpers.protocol_vmc_worker``.
"""

import time

import numpy as np

from .energy import KEYS, EnergyAccumulator


def device_of(wf):
    """The device handle a wave function lives on: the shared one of a ``MultiplyWF``, or a bare factor's own
    (``Slater`` / ``JastrowSpin`` / ``ThreeBodyJastrow`` built by ``generate_wf`` / ``generate_jastrow``); None otherwise."""
    dev = wf.fused_device() if hasattr(wf, "fused_device") else getattr(wf, "_dev", None)
    return dev if hasattr(dev, "vmc_sweeps") else None


def _fetch(dev, configs):
    """Walkers of the device handle into the host container (periodic: folded positions + wrap counters)."""
    if getattr(dev, "twisted", False):  # the handle keeps true (unfolded) coordinates: fold them back into the container
        from .configs import enforce_pbc

        configs.configs[...], configs.wrap[...] = enforce_pbc(configs.lvecs, dev.configs())
    else:
        configs.configs[...] = dev.configs()
        if dev.pbc:  # walkers stay folded into the cell; their wrap counters advance (coord.py:180-189)
            configs.wrap += dev.wrap_delta()


def _vmc_worker_host_accumulators(dev, wf, configs, tstep, nsteps, accumulators, tapes, seed, state_current):
    """Device sweeps, host accumulators: after every sweep (one ``pqa_vmc_sweeps`` call) the walkers come back and each
    accumulator's ``avg(configs, wf)`` runs as in mc.py:142-148 — through the protocol entry points, on the state the sweep left."""
    if seed is None:
        seed = int(np.random.randint(0, 2**31 - 1))
    # one Philox key per sweep, seed * nsteps + step: `vmc` hands block b the seed s + b, and keys s + b + step would give sweep
    # k + 1 of block b the draws of sweep k of block b + 1
    if not state_current:
        wf.recompute(configs)
    block_avg = {}
    t_move = t_acc = 0.0
    acc_sum = 0.0
    for step in range(nsteps):
        t0 = time.perf_counter()
        g, u = tapes.get("gauss"), tapes.get("unif")
        acc, _, rec = dev.vmc_sweeps(tstep, 1, gauss=None if g is None else g[step : step + 1], unif=None if u is None else u[step : step + 1],
                                     seed=seed * nsteps + step, energy=False, record="record" in tapes)
        if "record" in tapes:
            tapes["record"].append(rec)
        _fetch(dev, configs)
        t1 = time.perf_counter()
        for k, accumulator in accumulators.items():
            for m, res in accumulator.avg(configs, wf).items():
                block_avg[k + m] = block_avg.get(k + m, 0.0) + res / nsteps
        t_move, t_acc, acc_sum = t_move + (t1 - t0), t_acc + (time.perf_counter() - t1), acc[-1]
    block_avg["acceptance"] = acc_sum
    block_avg["move time"] = t_move / nsteps
    block_avg["accumulator time"] = t_acc / nsteps
    return block_avg, configs


def vmc_worker(wf, configs, tstep, nsteps, accumulators, tapes=None, seed=None, state_current=False, fetch_configs=True):
    """One block of ``nsteps`` sweeps (mc.py:102-153): returns (block averages, configs).

    ``state_current`` / ``fetch_configs`` are for block loops on the fused device path (``vmc`` below): the reference's worker
    is a function of (wf, configs) and so recomputes the wave function from ``configs`` on entry and hands the walkers back
    on exit — 100 MB each way over PCIe plus a full recompute per block at 65536 walkers of the 64-electron system.  A caller
    that knows the device already holds the state of ``configs`` (it ran the previous block and touched nothing since) passes
    ``state_current=True``; one that does not need the walkers on the host after this block passes ``fetch_configs=False``
    (open systems only: periodic containers also carry the block's wrap counters)."""
    dev = device_of(wf)
    if dev is None:
        raise TypeError("pyqmc_amd.vmc_worker drives a wave function that lives on one device handle (generate_wf); for anything else "
                        "run the reference's own pyqmc.method.mc.vmc_worker over the protocol objects (INTEGRATION.md)")
    if not all(isinstance(a, EnergyAccumulator) for a in accumulators.values()):
        return _vmc_worker_host_accumulators(dev, wf, configs, tstep, nsteps, accumulators, tapes or {}, seed, state_current)
    tapes = tapes or {}
    if seed is None:
        seed = int(np.random.randint(0, 2**31 - 1))
    if not state_current:
        wf.recompute(configs)
    block_avg = {}
    thr = next(iter(accumulators.values())).threshold if accumulators else 10.0
    if accumulators:
        next(iter(accumulators.values())).bind(dev)
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
    _fetch(dev, configs)
    return block_avg, configs


def vmc(wf, configs, nblocks=10, nsteps_per_block=10, tstep=0.5, accumulators=None, verbose=False, seed=None,
        hdf_file=None, continue_from=None, recompute_every=10, worker=None):
    """Block loop of ``pyqmc.method.mc.vmc`` (mc.py:176-274): returns (dict of arrays over the blocks run, configs).
    ``hdf_file``: block output in the reference's on-disk layout (``pyqmc_amd.blockfile``: HDF5 when h5py exists, NumPy
    archives otherwise); an existing file — or ``continue_from`` — restarts from its walkers at ``block[-1] + 1``, and
    ``nblocks`` counts the blocks of all calls together, as in the reference (mc.py:223-243).  ``worker``: a
    ``(wf, configs, tstep, nsteps, accumulators) -> (block, configs)`` callable to run the blocks with instead of the device
    worker (the test suite passes a protocol-route worker for the CPU oracle's wave functions)."""
    from .blockfile import BlockFile

    accumulators = accumulators or {}
    recompute_every = max(int(recompute_every), 1)
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
    dev = device_of(wf) if worker is None and all(isinstance(a, EnergyAccumulator) for a in accumulators.values()) else None
    current = False  # the device holds the wave-function state of the walkers it moved in the previous block
    for block in range(first, nblocks):
        # fused path: the walkers stay on the device from block to block; the state is rebuilt from the (fetched) walkers every
        # `recompute_every` blocks to bound the Sherman-Morrison round-off, and the host copy is refreshed when a block is
        # written to disk, before such a rebuild, and at the end
        rebuild_next = dev is not None and (block + 1 - first) % recompute_every == 0
        fetch = out is not None or block == nblocks - 1 or rebuild_next
        if worker is not None:
            blk, configs = worker(wf, configs, tstep, nsteps_per_block, accumulators)
        else:
            blk, configs = vmc_worker(wf, configs, tstep, nsteps_per_block, accumulators,
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
