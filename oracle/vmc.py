"""Oracle: single-electron-move VMC sweep (test infrastructure).

Follows ``vmc_worker`` (``pyqmc/method/mc.py:102-153``) and ``limdrift`` (:76-89),
keeping the reference's structure: two ``gradient_value`` calls per proposed move
(old and new position), host accept/reject, ``updateinternals(mask=accept)``, and an
energy evaluation after every sweep.  Random numbers come from explicit tapes:

  gauss (nsteps, N, W, 3)  standard normals (scaled by sqrt(tstep) here),
  unif  (nsteps, N, W)     Metropolis uniforms,
  ecp_rot (nsteps, N, n_ecp_atoms, 3, 3), ecp_unif (nsteps, N, n_ecp_atoms, W).
"""

import time

import numpy as np

from . import energy as oenergy


def limdrift(g, cutoff=1.0):
    tot = np.linalg.norm(g, axis=1)
    big = tot > cutoff
    with np.errstate(divide="ignore", invalid="ignore"):
        scaled = cutoff * g / tot[:, None]
    return np.where(big[:, None], scaled, g)


def vmc_worker(mol, wf, configs, tstep, gauss, unif, ecp_rot=None, ecp_unif=None, threshold=10.0,
               with_energy=True, record=None, ewald_kws=None, margins=None):
    """Returns (block_avg dict, configs).  ``record`` (optional list) receives the
    per-move accept masks for trajectory comparison, ``margins`` (optional list) the per-move ``ratio - u`` of the
    Metropolis test (how far each decision was from flipping)."""
    nsteps = gauss.shape[0]
    W, N, _ = configs.configs.shape
    block_avg = {}
    wf.recompute(configs)
    sq = np.sqrt(tstep)
    for step in range(nsteps):
        acc = 0.0
        t0 = time.perf_counter()
        for e in range(N):
            g, _, _ = wf.gradient_value(e, configs.electron(e))
            grad = limdrift(np.real(g.T))
            gs = gauss[step, e] * sq
            newpos = configs.make_irreducible(e, configs.configs[:, e, :] + gs + grad * tstep)
            g, new_val, saved = wf.gradient_value(e, newpos)
            new_grad = limdrift(np.real(g.T))
            forward = np.sum(gs**2, axis=1)
            backward = np.sum((gs + tstep * (grad + new_grad)) ** 2, axis=1)
            t_prob = np.exp(1.0 / (2.0 * tstep) * (forward - backward))
            ratio = np.abs(new_val) ** 2 * t_prob
            accept = ratio > unif[step, e]
            configs.move(e, newpos, accept)
            wf.updateinternals(e, newpos, configs, mask=accept, saved_values=saved)
            acc += np.mean(accept) / N
            if record is not None:
                record.append(accept.copy())
            if margins is not None:
                margins.append(ratio - unif[step, e])
        t1 = time.perf_counter()
        if with_energy:
            en = oenergy.energy(mol, configs, wf, threshold,
                                None if ecp_rot is None else ecp_rot[step],
                                None if ecp_unif is None else ecp_unif[step], ewald_kws=ewald_kws)
            for k, v in en.items():
                block_avg["energy" + k] = block_avg.get("energy" + k, 0.0) + np.mean(v, axis=0) / nsteps
        t2 = time.perf_counter()
        block_avg["acceptance"] = acc
        block_avg["move time"] = t1 - t0
        block_avg["accumulator time"] = t2 - t1
    return block_avg, configs
