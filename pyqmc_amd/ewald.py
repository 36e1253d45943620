"""Host set-up of the Ewald sum the device evaluates (``pyqmc/observables/ewald.py``).

Everything here is position independent: the partition parameter ``alpha = 5 / smallest cell height`` (:139-141),
the reciprocal vectors of the positive half space whose weight ``4 pi exp(-G^2 / 4 alpha^2) / (V G^2)`` exceeds 1e-10
(:143-146, :372-388), the ionic structure factor (:233-234), the ion-ion energy (:192-238) and the self +
charged-system constants (:150-190).  The walker-dependent sums run in ``k_ewald`` (csrc/pqa_energy.hpp).
"""

import numpy as np
from scipy.special import erfc

from .configs import MinimalImageDistance


def positive_gpoints(gmax, recvec, alpha, cellvolume):
    """G vectors (rows) and weights, in the order ``generate_positive_gpoints`` + ``select_big`` produce them
    (:372-388).  Only the index box that can hold weights above 1e-10 is enumerated (the reference filters a
    (gmax, 2 gmax+1, 2 gmax+1) grid; the survivors and their order are the same)."""
    lo, hi = 1e-8, 1e8  # bisect G^2 where the weight crosses 1e-10 (monotone decreasing)
    f = lambda g2: 4 * np.pi * np.exp(-g2 / (4 * alpha**2)) / (cellvolume * g2)
    for _ in range(200):
        mid = np.sqrt(lo * hi)
        lo, hi = (mid, hi) if f(mid) > 1e-10 else (lo, mid)
    gnorm = np.sqrt(hi) * (1 + 1e-9)
    latvec = np.linalg.inv(recvec).T  # recvec = inv(latvec).T
    n = np.minimum(np.floor(gnorm * np.linalg.norm(latvec, axis=1) / (2 * np.pi)).astype(int) + 1, gmax)
    blocks = [np.mgrid[1 : n[0] + 1, -n[1] : n[1] + 1, -n[2] : n[2] + 1].reshape(3, -1),
              np.mgrid[0:1, 1 : n[1] + 1, -n[2] : n[2] + 1].reshape(3, -1),
              np.mgrid[0:1, 0:1, 1 : n[2] + 1].reshape(3, -1)]
    gpts = np.concatenate(blocks, axis=1)
    gpoints = np.einsum("ji,jk->ik", gpts, recvec * 2 * np.pi)
    g2 = np.einsum("jk,jk->j", gpoints, gpoints)
    gweight = 4 * np.pi * np.exp(-g2 / (4 * alpha**2)) / (cellvolume * g2)
    big = gweight > 1e-10
    return gpoints[big], gweight[big], np.ascontiguousarray(gpts.T[big], dtype=np.int32)


def ewald_tables(cell, ewald_gmax=200, nlatvec=1):
    if nlatvec != 1:
        raise NotImplementedError("the device real-space sum runs over the 27 cells of nlatvec = 1 (the reference's default)")
    latvec = np.asarray(cell.lattice_vectors(), dtype=float)
    charges = np.asarray(cell.atom_charges(), dtype=float)
    coords = np.asarray(cell.atom_coords(), dtype=float)
    ne = int(np.sum(cell.nelec))
    vol = np.linalg.det(latvec)
    recvec = np.linalg.inv(latvec).T
    alpha = 5.0 / np.amin(1 / np.linalg.norm(recvec, axis=1))
    gpoints, gweight, gidx = positive_gpoints(ewald_gmax, recvec, alpha, vol)
    i_sum, ii_sum2 = charges.sum(), np.sum(charges**2)
    ii_sum = (i_sum**2 - ii_sum2) / 2
    ijconst = -np.pi / (vol * alpha**2)
    squareconst = -alpha / np.sqrt(np.pi) + ijconst / 2
    ii_const = ii_sum * ijconst + ii_sum2 * squareconst
    # ion-ion (:192-238)
    if len(charges) == 1:
        real = 0.0
    else:
        xyz = np.stack(np.meshgrid(*[np.arange(-1, 2)] * 3, indexing="ij"), axis=-1).reshape(-1, 3)
        disp = xyz @ latvec
        d, ij = MinimalImageDistance(latvec).dist_matrix(coords[np.newaxis])
        r = np.linalg.norm(d[:, :, np.newaxis, :] + disp, axis=-1)
        qq = np.prod(charges[np.asarray(ij)], axis=1)
        real = np.einsum("j,ijk->", qq, erfc(alpha * r) / r)
    ion_exp = np.exp(1j * gpoints @ coords.T) @ charges
    ion_ion = real + gweight @ np.abs(ion_exp) ** 2
    return {
        "alpha": float(alpha), "gpoints": np.ascontiguousarray(gpoints), "gweight": np.ascontiguousarray(gweight),
        "gidx": gidx, "recip": np.ascontiguousarray(recvec * 2 * np.pi),  # gpoints == gidx @ recip
        "ion_cos": np.ascontiguousarray(ion_exp.real), "ion_sin": np.ascontiguousarray(ion_exp.imag),
        "ee_const": float(ne * (ne - 1) / 2 * ijconst + ne * squareconst),
        "ei_const": float(-ne * i_sum * ijconst),
        "ii": float(ion_ion + ii_const),
    }
