"""Oracle: the batched ECP integrator (test infrastructure) — ``pyqmc/observables/jax_ecp.py``.

``evaluate_vl`` :160-222 (one table of all atoms' quadrature points per electron; v_l(r) (2l+1) P_l w_i per channel; the point's
probability = sum of the atom's non-local v_l^2), ``downselect_move_info`` :225-290 (top ``nsd`` kept with weight 1, ``nsr``
sampled from the renormalised rest through the cumulative sum, weight 1/(nsr p)), ``ECPAccumulator.__call__`` :72-105,
``nonlocal_tmoves`` :110-135.  The sort is STABLE here (numpy's default, which the reference calls, is not: ties between the
points of one atom are then ordered by the sort implementation).  Randomness is injected: ``rot`` one rotation per (electron, ECP
atom), ``unif`` (N, W, nsr).
"""

import numpy as np

from .energy import ecp_atoms, ecp_channels, legendre, quadrature, v_l


def default_naip(mol):
    """jax_ecp.py:43-54, for the atoms that carry an ECP."""
    out = []
    for ia in ecp_atoms(mol):
        max_l = max(int(l) for l, _ in mol._ecp[mol.atom_pure_symbol(ia)][1])
        out.append({0: 6, 1: 6, 2: 12}.get(max_l, 0))
    return np.asarray(out, dtype=int)


def evaluate_vl(mol, configs, e, naip, rot):
    """jax_ecp.py:160-222.  Returns (r_ea_vec, r_ea_i, prob, v_l, P_l, local); rot (n_ecp_atoms, 3, 3)."""
    x = configs.configs
    W = x.shape[0]
    atoms = ecp_atoms(mol)
    chans = [ecp_channels(mol._ecp[mol.atom_pure_symbol(ia)]) for ia in atoms]
    maxl = max(len(c) for c in chans)
    npts = int(np.sum(naip))
    vl, pl = np.zeros((W, npts, maxl)), np.zeros((W, npts, maxl))
    r_i, prob, r_vec, local = np.zeros((W, npts, 3)), np.zeros((W, npts)), np.zeros((W, npts, 3)), np.zeros(W)
    beg = 0
    for k, ia in enumerate(atoms):
        d = x[:, e, :] - np.asarray(mol.atom_coords()[ia])
        if hasattr(mol, "a"):
            from .pbc import minimal_image

            d = minimal_image(mol.lattice_vectors())(d)
        r = np.linalg.norm(d, axis=-1)
        v = v_l(chans[k], r)  # non-local channels first, local last
        nl = v.shape[1]
        local += v[:, -1]
        n = int(naip[k])
        if n == 0:
            continue
        pts, wts = quadrature(n)
        ri = r[:, None, None] * (rot[k] @ pts.T).T[None]
        cos = np.einsum("ik,ijk->ij", d, ri) / (r[:, None] * np.linalg.norm(ri, axis=-1))
        P = np.zeros((W, n, nl))
        for c in range(nl - 1):
            P[:, :, c] = (2 * c + 1) * legendre(cos, c) * wts[None]
        sl = slice(beg, beg + n)
        vl[:, sl, :nl] = v[:, None, :] * P
        pl[:, sl, :nl] = P
        r_i[:, sl] = ri
        prob[:, sl] = np.sum(v[:, :-1] ** 2, axis=-1)[:, None]
        r_vec[:, sl] = d[:, None]
        beg += n
    return r_vec, r_i, prob, vl, pl, local


def downselect(info, nsd, nsr, unif):
    """jax_ecp.py:225-290; unif (W, nsr)."""
    r_vec, r_i, prob, vl, pl = info
    W, npts, _ = vl.shape
    if nsr + nsd >= npts:
        return info
    p = prob.copy()
    det = np.argsort(p, axis=1, kind="stable")[:, npts - nsd:]
    np.put_along_axis(p, det, 0.0, axis=1)
    norm = np.sum(p, axis=1)
    p[norm == 0, :] = 1.0 / (npts - nsd)
    norm[norm == 0] = 1.0
    p = p / norm[:, None]
    cdf = np.cumsum(p, axis=1)
    rnd = (unif[:, None, :] > cdf[:, :, None]).sum(axis=1)
    idx = np.concatenate((det, rnd), axis=1)
    np.put_along_axis(p, det, 1.0 / nsr if nsr else 1.0, axis=1)
    psel = (nsr if nsr else 1.0) * np.take_along_axis(p, idx, axis=1)
    take = lambda a: np.take_along_axis(a, idx[:, :, None], axis=1)  # noqa: E731
    return take(r_vec), take(r_i), np.take_along_axis(prob, idx, axis=1), take(vl) / psel[:, :, None], take(pl)


def selected(mol, configs, e, naip, nsd, nsr, rot, unif):
    info = evaluate_vl(mol, configs, e, naip, rot)
    r_vec, r_i, prob, vl, pl = downselect(info[:5], nsd, nsr, unif)
    epos = (configs.configs[:, e, None, :] - r_vec) + r_i
    return epos, vl, pl, info[5], info[2]


def ecp(mol, configs, wf, naip, nsd, nsr, rot_tape, unif_tape):
    """ECPAccumulator.__call__ (jax_ecp.py:72-105).  rot_tape (N, n_ecp_atoms, 3, 3), unif_tape (N, W, nsr)."""
    W, N = configs.configs.shape[:2]
    tot = np.zeros(W, dtype=getattr(wf, "dtype", float))
    for e in range(N):
        epos, vl, pl, local, prob = selected(mol, configs, e, naip, nsd, nsr, rot_tape[e], unif_tape[e])
        tot += local
        if prob.sum() == 0:
            continue
        ratio = wf.testvalue(e, configs.make_irreducible(e, epos))[0]
        tot += np.einsum("na,nal->n", ratio, vl[:, :, :-1])
    return tot


def tmoves(mol, configs, wf, e, tau, naip, nsd, nsr, rot, unif):
    """ECPAccumulator.nonlocal_tmoves (jax_ecp.py:110-135)."""
    epos, vl, pl, _, _ = selected(mol, configs, e, naip, nsd, nsr, rot, unif)
    ratio = np.asarray(wf.testvalue(e, configs.make_irreducible(e, epos))[0])
    ew = np.zeros_like(pl)
    m = pl > 0
    ew[m] = np.exp(-tau * vl[m] / pl[m]) - 1
    return {"ratio": ratio, "weight": np.einsum("ijk,ijk->ij", ew, pl), "epos": epos}
