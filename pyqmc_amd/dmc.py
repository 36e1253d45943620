"""Diffusion Monte Carlo drivers — counterpart of ``pyqmc/method/dmc.py``.

The step loop (T-moves, drift-diffusion with fixed-node rejection, weights: dmc.py:123-221) runs on the device
(``pqa_dmc_steps``, csrc/pqa_dmc.hpp); here are its host wrapper, the stochastic comb (``branch`` dmc.py:342-376: an
index computation on W weights) and the block loop of ``rundmc`` (:413-591) with restart files.  The reference's
per-electron host loop is not restated: the parity tests drive the protocol entry points through
``tests/helpers.protocol_dmc_propagate``, and an unmodified ``pyqmc.method.dmc`` runs over the wave-function objects.

``rng`` (optional, tests) supplies the random draws to replay the reference's:
``normal(W)->(W,3)``, ``rand(W)->(W,)``, ``rand1()->float``, ``rot()->(3,3)``, ``random(W)->(W,)``.
"""

import numpy as np


def _record_tapes(rng, nsteps, N, necp, W, tmoves):
    """Draw everything ``nsteps`` steps consume from ``rng`` in the reference's order (dmc.py:146-196) and lay it out as
    the replay tapes of ``pqa_dmc_steps``."""
    t = {"gauss": np.empty((nsteps, N, W, 3)), "unif": np.empty((nsteps, N, W))}
    if necp:
        t["ecp_rot"], t["ecp_unif"] = np.empty((nsteps + 1, N, necp, 3, 3)), np.empty((nsteps + 1, N, necp, W))
    if tmoves:
        t["tm_rot"], t["tm_unif"] = np.empty((nsteps, N, necp, 3, 3)), np.empty((nsteps, N, necp, W))
        t["tm_u1"], t["tm_u2"] = np.empty((nsteps, N, W)), np.empty((nsteps, N, W))

    def energy_draws(i):
        for e in range(N):
            for k in range(necp):
                t["ecp_unif"][i, e, k] = rng.random(W)
                t["ecp_rot"][i, e, k] = rng.rot()

    energy_draws(0)
    for i in range(nsteps):
        if tmoves:
            for e in range(N):
                for k in range(necp):
                    t["tm_unif"][i, e, k] = rng.random(W)
                    t["tm_rot"][i, e, k] = rng.rot()
                t["tm_u1"][i, e] = [rng.rand1() for _ in range(W)]
                t["tm_u2"][i, e] = rng.rand(W)
        for e in range(N):
            t["gauss"][i, e] = rng.normal(W)
            t["unif"][i, e] = rng.rand(W)
        energy_draws(i + 1)
    return t


def _refuse_batched_tmoves(acc, necp):
    """Both device step loops (all-device and one step per call) draw their T-move candidates inside ``pqa_dmc_steps`` from the semi-local
    integrator's table (eval_ecp.compute_tmoves); an accumulator bound with ``use_old_ecp=False`` would have its ENERGY from the batched
    integrator and its T-MOVES from the other one — refused in both, not mixed silently."""
    if acc.has_nonlocal_moves() and necp > 0 and not getattr(acc, "use_old_ecp", True):
        raise NotImplementedError("the device DMC driver draws its T-move candidates from the semi-local integrator's table (eval_ecp.compute_tmoves); "
                                  "with use_old_ecp=False run the reference's pyqmc.method.dmc over the protocol objects: "
                                  "EnergyAccumulator.nonlocal_tmoves then serves the batched candidates (INTEGRATION.md)")


def _propagate_fused(dev, wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps, acc, name, rng, state_current=False):
    """``dmc_propagate`` through ``pqa_dmc_steps``: the whole step loop stays on the device."""
    from .energy import KEYS

    W, N = configs.configs.shape[:2]
    necp = getattr(dev, "necp", 0)
    tmoves = acc.has_nonlocal_moves() and necp > 0
    _refuse_batched_tmoves(acc, necp)
    if not state_current:  # the reference recomputes at the start of every block (dmc.py:155)
        wf.recompute(configs)
    acc.bind(dev)
    tapes = None if rng is None else _record_tapes(rng, nsteps, N, necp, W, tmoves)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    # like vmc_worker: the device Philox streams are keyed by a seed drawn from numpy's global generator, so
    # numpy.random.seed() controls reproducibility and ranks that seed numpy differently get independent streams
    avg, stat = dev.dmc_steps(tstep, nsteps, w, branchcut, e_trial, e_est, threshold=acc.threshold, tapes=tapes,
                              seed=int(np.random.randint(0, 2**31 - 1)))
    weights[:] = w
    from .vmc import _fetch

    _fetch(dev, configs)  # (periodic: folded positions + wrap counters; twisted handles keep true coordinates on the device)
    wts = avg[:, 6]
    rel = wts / wts.mean()
    out = {name + k: np.mean(avg[:, i] * rel) for i, k in enumerate(KEYS[:6])}
    if avg.shape[1] > 7:  # complex wave function: ecp and total carry an imaginary part (eval_ecp.py:89), the other keys are real
        im = np.mean(avg[:, 7] * rel)
        for k in ("ecp", "total"):
            out[name + k] = complex(out[name + k], im)
    out["acceptance"] = np.mean(stat[:, 0] * rel)
    out["tmove_acceptance"] = np.mean(stat[:, 1] * rel)
    out["weight"] = wts.mean()
    return out, configs, weights


def _propagate_host_accumulators(dev, wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps, accumulators, ekey, state_current=False,
                                 rng=None):
    """``dmc_propagate`` with further accumulators (density matrices, ...) next to the energy: every step is one device call
    (``pqa_dmc_steps`` with ``nsteps = 1``), after which the walkers come back and the other accumulators run on the host over
    the protocol entry points, weight-averaged as in dmc.py:205-221.  From the second step on the device starts from the
    energies its previous step ended with (``pqa_dmc_continue``), as the reference carries ``eloc`` / ``v2`` from step to step
    (dmc.py:148-149, :199-200): same trajectory as the all-device loop, one energy evaluation per step.  ``rng`` replays the
    reference's draws (tests)."""
    from .energy import KEYS
    from .vmc import _fetch

    acc, name = accumulators[ekey[0]], ekey[0]
    W = configs.configs.shape[0]
    _refuse_batched_tmoves(acc, getattr(dev, "necp", 0))
    if not state_current:
        wf.recompute(configs)
    acc.bind(dev)
    w = np.ascontiguousarray(weights, dtype=np.float64)
    df = []
    tapes = None
    if rng is not None:
        N, necp = configs.configs.shape[1], getattr(dev, "necp", 0)
        tapes = _record_tapes(rng, nsteps, N, necp, W, acc.has_nonlocal_moves() and necp > 0)
    for i in range(nsteps):
        step_tapes = None
        if tapes is not None:  # this step's slice; the energy draws: [starting configuration (used by the first call only), this step]
            step_tapes = {k: (np.stack([v[0 if i == 0 else i], v[i + 1]]) if k.startswith("ecp_") else v[i:i + 1]) for k, v in tapes.items()}
        # a host accumulator of the previous step may have touched the device state (wf.recompute, parameter setters): the energies the
        # device kept are then gone and the step starts from a fresh evaluation, as the first one does
        avg, stat = dev.dmc_steps(tstep, 1, w, branchcut, e_trial, e_est, threshold=acc.threshold, tapes=step_tapes,
                                  seed=int(np.random.randint(0, 2**31 - 1)), cont=i > 0 and dev.dmc_can_continue())
        _fetch(dev, configs)
        wavg = float(avg[0, 6])
        d = {name + k: avg[0, i] for i, k in enumerate(KEYS[:6])}
        if avg.shape[1] > 7:
            for k in ("ecp", "total"):
                d[name + k] = complex(d[name + k], avg[0, 7])
        for k, other in accumulators.items():
            if k == name:
                continue
            for m, res in other(configs, wf).items():
                d[k + m] = np.einsum("...i,i...->...", w, res) / (W * wavg)
        d["weight"], d["acceptance"], d["tmove_acceptance"] = wavg, stat[0, 0], stat[0, 1]
        df.append(d)
    weights[:] = w
    wts = np.asarray([d["weight"] for d in df])
    rel = wts / wts.mean()
    out = {k: np.mean([d[k] * r for d, r in zip(df, rel)], axis=0) for k in df[0]}
    out["weight"] = wts.mean()
    return out, configs, weights


def fused_dmc_supported(wf, accumulators, ekey):
    """The device step loop covers wave functions on one handle (real or complex) whose energy accumulator is ours; further
    accumulators run on the host between device steps (``_propagate_host_accumulators``)."""
    from .energy import EnergyAccumulator
    from .vmc import device_of

    dev = device_of(wf)
    if dev is None or ekey[0] not in accumulators or ekey[1] != "total":
        return None
    return dev if isinstance(accumulators[ekey[0]], EnergyAccumulator) else None


def dmc_propagate(wf, configs, weights, tstep, branchcut_start, e_trial, e_est, nsteps=5, accumulators=None,
                  ekey=("energy", "total"), rng=None, state_current=False):
    """Propagate ``nsteps`` DMC steps without branching; returns (block averages, configs, weights) with the
    reference's keys (``<acc><quantity>``, ``weight``, ``acceptance``, ``tmove_acceptance``).

    The whole step loop of dmc.py:123-221 runs on the device (``pqa_dmc_steps``): wave functions on one handle — real, or
    complex (no node constraint, weights from Re E_L, T-move amplitudes from Re[Psi(R')/Psi(R)]: golden g30) — with the
    energy accumulator; further accumulators (OBDM, TBDM, ...) are evaluated on the host between device steps and
    weight-averaged as dmc.py:205-212 does.  ``rng`` replays the reference's draws (tests); ``state_current=True``
    promises that the device already holds the wave-function state of ``configs`` — ``rundmc`` passes it after branching on
    the device (``DeviceWF.resample``) — and skips the initial recompute.  (``tests/helpers.protocol_dmc_propagate`` is the
    protocol-route harness the parity tests use.)"""
    assert accumulators is not None, "Need an energy accumulator for DMC"
    dev = fused_dmc_supported(wf, accumulators, ekey)
    if dev is None:
        raise NotImplementedError("pyqmc_amd.dmc_propagate runs wave functions on one device handle whose energy comes from "
                                  "pyqmc_amd.EnergyAccumulator; drive pyqmc.method.dmc.dmc_propagate over the protocol objects for anything else")
    if set(accumulators) != {ekey[0]}:
        return _propagate_host_accumulators(dev, wf, configs, weights, tstep, branchcut_start, e_trial, e_est, nsteps, accumulators, ekey,
                                            state_current=state_current, rng=rng)
    return _propagate_fused(dev, wf, configs, weights, tstep, branchcut_start, e_trial, e_est, nsteps, accumulators[ekey[0]], ekey[0], rng,
                            state_current=state_current)


def comb_indices(weights, base_u):
    """Stochastic comb: resampling indices for walkers with the given weights (global order)."""
    W = len(weights)
    cum = np.cumsum(weights)
    wtot = cum[-1]
    teeth = (base_u * wtot + np.linspace(0, wtot, W, endpoint=False)) % wtot
    return np.searchsorted(cum, teeth), wtot


def branch(configs, weights, base_u=None, on_resample=None):
    """Single-process branching: walkers resampled in proportion to their weights, weights reset to the mean.
    ``on_resample(newinds)`` lets the wave-function state follow the walkers (``DeviceWF.resample``)."""
    base_u = np.random.rand() if base_u is None else base_u
    newinds, wtot = comb_indices(weights, base_u)
    unique, counts = np.unique(newinds, return_counts=True)
    configs.resample(newinds)
    if on_resample is not None:
        on_resample(newinds)
    weights.fill(wtot / len(weights))
    return configs, weights, {"max branches": int(counts.max()), "Number of walkers killed": int(len(weights) - len(unique))}


def rundmc(wf, configs, weights=None, tstep=0.01, nblocks=10, nsteps_per_block=None, accumulators=None, verbose=False,
           ekey=("energy", "total"), vmc_warmup=10, branchcut_start=10, feedback=1.0, distributed=False, recompute_every=10,
           hdf_file=None, continue_from=None, blockoffset=0, propagate=None, vmc_worker=None):
    """Block loop of the reference's ``rundmc`` (dmc.py:413-591): VMC warm-up and energy reference — or, when ``hdf_file``
    exists / ``continue_from`` is given, the walkers, weights, ``e_trial``, ``e_est``, ``esigma`` and block offset of that
    file (dmc.py:466-500) — then propagate -> branch -> trial-energy feedback per block.  With ``distributed=True`` every
    rank calls this with its own walker shard and the energy sums / branching go through ``pyqmc_amd.dist`` (RCCL or gloo).

    Runs on the fused path branch ON THE DEVICE: the comb's indices gather the wave-function state (``pqa_resample``;
    sharded runs: ``dist.branch_distributed`` -> ``pqa_branch_exchange``, which also recomputes the walkers that arrived
    from other ranks) instead of recomputing everything from the resampled coordinates as the reference does after every
    branch (dmc.py:155); a full recompute every ``recompute_every`` blocks bounds the round-off the Sherman-Morrison
    updates accumulate (``recompute_every=1`` is the reference's schedule).
    ``hdf_file``: per-block output, walkers and weights in the reference's on-disk layout (``dmc_file`` dmc.py:379-391;
    ``pyqmc_amd.blockfile``).  A sharded run writes one file per rank — rank 0 ``hdf_file`` itself, rank r
    ``hdf_file + ".rank<r>"`` — each with the (identical, all-reduced) block record and that rank's own walkers and
    weights, so no two processes ever touch one file and every rank can continue from its own.
    ``propagate`` / ``vmc_worker``: callables with the signatures of ``dmc_propagate`` / ``vmc.vmc_worker`` to run the blocks
    with instead of the device drivers (the CPU tests pass protocol-route drivers for the oracle's wave functions)."""
    from . import dist as pdist
    from .blockfile import BlockFile
    from .vmc import vmc

    def rank_path(path):
        if path is None or not distributed:
            return path
        import torch.distributed as td

        r = td.get_rank() if td.is_available() and td.is_initialized() else 0
        return path if r == 0 else f"{path}.rank{r}"

    hdf_file, continue_from = rank_path(hdf_file), rank_path(continue_from)
    out = None if hdf_file is None else BlockFile(hdf_file)
    nsteps_per_block = max(1, int(0.1 / tstep)) if nsteps_per_block is None else nsteps_per_block
    acc = accumulators[ekey[0]]
    if continue_from is not None and out is not None and out.exists():  # dmc.py:467-470
        raise RuntimeError(f"continue_from is set but hdf_file={hdf_file} already exists! Delete or rename {hdf_file} and try again.")
    if continue_from is None and out is not None and out.exists():
        continue_from = hdf_file
    history = {}
    if continue_from is not None:
        src = BlockFile(continue_from)
        if not src.exists():
            raise FileNotFoundError(f"continue_from={continue_from}: no such block file")
        data = src.datasets()
        if "e_trial" not in data:
            raise ValueError("Did not find e_trial in the restart file. This may mean that you are trying to restart from a different version of DMC")
        blockoffset = int(data["block"][-1]) + 1
        w_file = src.load_walkers(configs)
        weights = weights if w_file is None else w_file
        e_trial, e_est, esigma = float(np.real(data["e_trial"][-1])), float(np.real(data["e_est"][-1])), float(np.real(data["esigma"][-1]))
        if continue_from == hdf_file:  # estimate_energy reads the whole file (dmc.py:594-603): the earlier blocks count
            history = {"en": list(data[ekey[0] + ekey[1]]), "wt": list(data["weight"])}
        if verbose:
            print(f"Restarting calculation {continue_from} from block {blockoffset}")
    else:
        _, configs = vmc(wf, configs, nblocks=vmc_warmup, accumulators={}, verbose=verbose, worker=vmc_worker)
        wf.recompute(configs)
        en = np.real(acc(configs, wf)[ekey[1]])
        if distributed:
            (m1, m2), _ = pdist.allreduce_block([en.sum(), (en**2).sum()], len(en))
            eref, esigma = m1, np.sqrt(max(m2 - m1 * m1, 0.0))
        else:
            eref, esigma = en.mean(), en.std()
        e_trial = e_est = eref
    W = configs.configs.shape[0]
    weights = np.ones(W) if weights is None else weights
    rows = []
    en_hist, wt_hist = list(history.get("en", [])), list(history.get("wt", []))
    dev = None if propagate is not None else fused_dmc_supported(wf, accumulators, ekey)  # device-resident branching: single process AND sharded runs
    current = False
    for block in range(blockoffset, nblocks):
        if propagate is not None:
            blk, configs, weights = propagate(wf, configs, weights, tstep, branchcut_start * esigma, e_trial, e_est,
                                              nsteps=nsteps_per_block, accumulators=accumulators, ekey=ekey)
        else:
            blk, configs, weights = dmc_propagate(wf, configs, weights, tstep, branchcut_start * esigma, e_trial, e_est,
                                                  nsteps=nsteps_per_block, accumulators=accumulators, ekey=ekey,
                                                  state_current=current and block % max(int(recompute_every), 1) != 0)
        if distributed:  # weighted recombination of the per-rank block averages (dmc.py:238-304)
            keys = sorted(k for k in blk if k != "weight")
            parts = [np.asarray(blk[k] * blk["weight"] * W).ravel() for k in keys]  # (array-valued accumulators, complex energies)
            sums, _ = pdist.allreduce_block(np.concatenate(parts + [[blk["weight"] * W, W]]), 1)
            wsum, wtot_n = np.real(sums[-2]), np.real(sums[-1])
            offs = np.concatenate([[0], np.cumsum([len(p) for p in parts])])
            new = {}
            for k, a, b in zip(keys, offs[:-1], offs[1:]):
                v = (sums[a:b] / wsum).reshape(np.shape(blk[k]))
                new[k] = v if np.iscomplexobj(blk[k]) else np.real(v)
                new[k] = new[k][()] if np.ndim(new[k]) == 0 else new[k]
            blk = new
            blk["weight"] = wsum / wtot_n
            # weights all-gathered, identical comb on every rank, only re-assigned walkers exchanged (device buffers under
            # RCCL); the state of walkers that stay is gathered on the device, arrivals alone are recomputed
            configs, weights, info, wstd = pdist.branch_distributed(configs, weights, dev=dev)
            current = dev is not None
            blk["weight_std"] = wstd
            mean_w = float(pdist.allreduce_block([weights.sum()], len(weights))[0][0])
        else:
            blk["weight_std"] = np.std(weights)
            configs, weights, info = branch(configs, weights, on_resample=None if dev is None else dev.resample)
            current = dev is not None
            mean_w = np.mean(weights)
        blk.update(info, e_trial=e_trial, e_est=e_est, block=block, esigma=esigma, tstep=tstep, nsteps_per_block=nsteps_per_block)
        rows.append(blk)
        if out is not None:
            out.append(blk, {}, configs, weights)  # the reference's DMC file has no attributes: tstep is a per-block dataset
        en_hist.append(blk[ekey[0] + ekey[1]])
        wt_hist.append(blk["weight"])
        en_b, wt_b = np.array(en_hist), np.array(wt_hist)
        warm = len(en_b) // 4
        e_est = float(np.real(np.average(en_b[warm:], weights=wt_b[warm:])))  # estimate_energy, dmc.py:594-603
        e_trial = e_est - feedback * float(np.real(np.log(mean_w)))
        if verbose:
            print(f"block {block}: E={blk[ekey[0] + ekey[1]]:.6f} e_trial={e_trial:.6f} e_est={e_est:.6f} sigma(w)={blk['weight_std']:.3g} {info}")
    return ({k: np.asarray([r[k] for r in rows]) for k in rows[0]} if rows else {}), configs, weights
