"""Oracle: diffusion Monte Carlo propagation and branching (test infrastructure).

Follows ``pyqmc/method/dmc.py``: ``limdrift`` :22-35 (Umrigar), ``propose_drift_diffusion`` :49-70
(fixed-node sign rejection for real wave functions), ``propose_tmoves`` :73-120 with
``eval_ecp.compute_tmoves`` (``pyqmc/observables/eval_ecp.py:43-80``), ``dmc_propagate`` :123-221,
``compute_S`` :224-235, ``branch`` :342-376 (stochastic comb).

Randomness comes from a ``tape`` object exposing the four draws the reference makes, in its call order:
``normal(W)`` -> (W,3) standard normals, ``rand(W)`` -> (W,) uniforms, ``rand1()`` -> scalar uniform,
``rot()`` -> 3x3, ``random(W)`` -> (W,).
"""

import numpy as np

from . import energy as oenergy


def limdrift(g, tau, acyrus=0.5):
    v2 = np.sum(g**2, axis=1)
    taueff = np.full(v2.shape, float(tau))
    m = v2 > 1e-8
    taueff[m] = (np.sqrt(1 + 2 * tau * acyrus * v2[m]) - 1) / (acyrus * v2[m])
    return g * taueff[:, None]


def select_tmoves(ratio, weight, pos, current, select_u):
    """dmc.py:73-120 after the candidates are known.  select_u: (W,) uniforms (one per walker, in walker order).
    -> newpos (W,3), move_selected (W,), acceptance (W,).
    Complex wave functions: the reference's lines order complex amplitudes with `>` / `<` and store 1 / ratio in a real
    array, which has no defined meaning (and raises under NumPy 2); the rule followed — pinned by golden g30, generated from
    the reference with the real part of the ratios handed to these very lines — is amplitudes from Re[Psi(R')/Psi(R)]."""
    ratio = np.real(ratio)
    amp = ratio * weight
    fwd = np.where(amp > 0, amp, 0.0)
    norm = 1.0 + fwd.sum(axis=1)
    cdf = np.cumsum(fwd / norm[:, None], axis=1)
    sel = np.array([np.searchsorted(cdf[w], select_u[w]) for w in range(len(cdf))])
    chosen = sel < amp.shape[1]
    newpos = current.copy()
    back = amp.copy()
    for w in np.nonzero(chosen)[0]:
        m = sel[w]
        newpos[w] = pos[w, m]
        rr = 1.0 / ratio[w, m]
        back[w] *= rr
        back[w, m] = rr * weight[w, m]
    back[back < 0] = 0.0
    acc = norm / (1.0 + back.sum(axis=1))
    acc[~chosen] = 0.0
    return newpos, chosen, acc


def compute_S(e_trial, e_est, branchcut, v2, tau, eloc, nelec):
    e_cut = e_est - eloc
    m = np.abs(e_cut) > branchcut
    e_cut[m] = branchcut * np.sign(e_cut[m])
    return e_trial - e_est + e_cut / np.sqrt(1 + (v2 * tau / nelec) ** 2)


def dmc_propagate(mol, wf, configs, weights, tstep, branchcut_start, e_trial, e_est, nsteps, tape, threshold=10.0,
                  record=None):
    """dmc.py:123-221 with the EnergyAccumulator as the only accumulator.  Returns (df, configs, weights)."""
    W, N = configs.configs.shape[:2]
    has_ecp = bool(mol._ecp)
    necp = len(oenergy.ecp_atoms(mol))

    def energy():
        # the reference draws one mask-uniform vector and one rotation per (electron, atom) ecp_ea call
        rots, unifs = np.zeros((N, necp, 3, 3)), np.zeros((N, necp, W))
        for e in range(N):
            for k in range(necp):
                unifs[e, k] = tape.random(W)
                rots[e, k] = tape.rot()
        return oenergy.energy(mol, configs, wf, threshold, rots, unifs)

    wf.recompute(configs)
    en = energy()
    eloc, v2 = en["total"].real, en["grad2"]
    df = []
    for _ in range(nsteps):
        r2_acc, r2_prop = np.zeros(W), np.zeros(W)
        prob_acc, tm_acc = np.zeros(W), np.zeros(W)
        if has_ecp:
            for e in range(N):
                ratio, weight, pos = compute_tmoves(mol, configs, wf, e, threshold, tstep, tape)
                sel_u = np.array([tape.rand1() for _ in range(W)])
                newpos, chosen, acc = select_tmoves(ratio, weight, pos, configs.configs[:, e, :], sel_u)
                accept = chosen & (acc > tape.rand(W))
                ep = configs.make_irreducible(e, newpos)
                configs.move(e, ep, accept)
                wf.updateinternals(e, ep, configs, mask=accept)
                tm_acc += accept / N
                if record is not None:
                    record.append(("t", e, accept.copy()))
        for e in range(N):
            grad = limdrift(np.real(wf.gradient(e, configs.electron(e)).T), tstep)
            gauss = tape.normal(W) * np.sqrt(tstep)
            ep = configs.make_irreducible(e, configs.configs[:, e, :] + gauss + grad)
            g, wfratio, saved = wf.gradient_value(e, ep)
            new_grad = limdrift(np.real(g.T), tstep)
            fwd = np.sum(gauss**2, axis=1)
            bwd = np.sum((gauss + grad + new_grad) ** 2, axis=1)
            ratio = np.abs(wfratio) ** 2 * np.exp(1 / (2 * tstep) * (fwd - bwd))
            if not np.iscomplexobj(wfratio):
                ratio = ratio * np.sign(wfratio)  # fixed node only for real wave functions (dmc.py:64-66)
            accept = ratio > tape.rand(W)
            r2 = np.sum((gauss + grad) ** 2, axis=1)
            configs.move(e, ep, accept)
            wf.updateinternals(e, ep, configs, mask=accept, saved_values=saved)
            r2_prop += r2
            r2_acc[accept] += r2[accept]
            prob_acc += accept / N
            if record is not None:
                record.append(("d", e, accept.copy()))
        eloc_old, v2_old = eloc.copy(), v2.copy()
        en = energy()
        eloc, v2 = en["total"].real, en["grad2"]
        tdamp = r2_acc / r2_prop
        Snew = compute_S(e_trial, e_est, branchcut_start, v2, tstep, eloc, N)
        Sold = compute_S(e_trial, e_est, branchcut_start, v2_old, tstep, eloc_old, N)
        weights = weights * np.exp(tstep * tdamp * (0.5 * Snew + 0.5 * Sold))
        wavg = np.mean(weights)
        avg = {"energy" + k: np.dot(weights, v) / (W * wavg) for k, v in en.items()}
        avg.update(weight=wavg, acceptance=np.mean(prob_acc), tmove_acceptance=np.mean(tm_acc))
        df.append(avg)
    wt = np.array([d["weight"] for d in df])
    aw = wt / np.mean(wt)
    ret = {k: np.mean([d[k] * w for d, w in zip(df, aw)], axis=0) for k in df[0]}
    ret["weight"] = np.mean(wt)
    return ret, configs, weights


def compute_tmoves(mol, configs, wf, e, threshold, tau, tape):
    """eval_ecp.compute_tmoves (eval_ecp.py:43-80): candidates over all ECP atoms' quadrature points.
    -> ratio (W,P), weight (W,P), pos (W,P,3); walkers failing the ECP mask get ratio 1, weight 0."""
    W = configs.configs.shape[0]
    ratios, weights, poss = [], [], []
    for ia in oenergy.ecp_atoms(mol):
        unif = tape.random(W)
        rot = tape.rot()
        d = oenergy.ecp_ea(mol, configs, wf, e, ia, threshold, rot, unif)
        npts = d["P_l"].shape[1]
        w = np.zeros((W, npts))
        r = np.ones((W, npts), dtype=np.asarray(d["ratio"]).dtype)
        w[d["mask"]] = np.einsum("ik,ijk->ij", np.exp(-tau * d["v_l"]) - 1, d["P_l"])
        r[d["mask"]] = d["ratio"]
        ratios.append(r)
        weights.append(w)
        poss.append(d["epos"])
    return np.concatenate(ratios, axis=1), np.concatenate(weights, axis=1), np.concatenate(poss, axis=1)


def branch(configs_array, weights, base_u):
    """dmc.py:342-376: stochastic comb.  -> (newinds, new weights, info)"""
    W = len(weights)
    prob = np.cumsum(weights)
    wtot = prob[-1]
    newinds = np.searchsorted(prob, (base_u * wtot + np.linspace(0, wtot, W, endpoint=False)) % wtot)
    unique, counts = np.unique(newinds, return_counts=True)
    return newinds, np.full(W, wtot / W), {"max branches": int(np.max(counts)), "Number of walkers killed": int(W - len(unique))}
