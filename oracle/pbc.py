"""TEST INFRASTRUCTURE — CPU restatement of the reference's periodic-boundary pieces (numpy, fp64).

* ``enforce_pbc``            pyqmc/pbc/pbc.py:18-49
* ``minimal_image``          pyqmc/configurations/distance.py:83-159 (diagonal / orthogonal / general rules)

Pinned by tests/golden/g13_pbc.npz (outputs of the real reference, tests/golden/make_golden.py:g_pbc), which also
holds the known-answer table of the reference's own tests/unit/test_pbcs.py:19-72.
"""

import numpy as np


def enforce_pbc(lattvecs, epos):
    frac = np.einsum("...ij,jk->...ik", epos, np.linalg.inv(lattvecs))
    wrap, rem = np.divmod(frac, 1)
    return np.dot(rem, lattvecs), wrap


def lattice_kind(latvec, tol=1e-10):
    def diag(m):
        return np.all(np.abs(m - np.diag(np.diagonal(m))) < tol)

    if diag(latvec):
        return "diagonal"
    return "orthogonal" if diag(latvec @ latvec.T) else "general"


def minimal_image(latvec):
    """Returns f(d) mapping displacement arrays (...,3) to their nearest-image representatives."""
    latvec = np.asarray(latvec, dtype=float)
    kind = lattice_kind(latvec)
    inv = np.linalg.inv(latvec)
    grid = np.meshgrid(*[np.arange(3)] * 3)
    shifts = (np.stack([g.ravel() for g in grid], axis=0).T - 1) @ latvec  # distance.py:113-118

    def f(d):
        d = np.array(d, dtype=float)
        if kind == "diagonal":
            for i in range(3):
                L = latvec[i, i]
                d[..., i] = (d[..., i] + L / 2) % L - L / 2
            return d
        if kind == "orthogonal":
            return (((d @ inv) + 0.5) % 1 - 0.5) @ latvec
        cand = d[None] + shifts.reshape((-1,) + (1,) * (d.ndim - 1) + (3,))
        best = np.argmin(np.sum(cand**2, axis=-1), axis=0)
        return np.take_along_axis(cand, best[None, ..., None], axis=0)[0]

    return f
