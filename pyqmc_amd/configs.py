"""Walker containers with the interface of ``pyqmc/configurations/coord.py:21-112``.

``OpenConfigs``/``OpenElectron`` (``coord.py:21-112``) and ``PeriodicConfigs``/``PeriodicElectron``
(``coord.py:115-252``) with ``enforce_pbc`` (``pyqmc/pbc/pbc.py:18-49``) and the minimal-image distance
(``distance.py:83-159``).  The arrays are ordinary host ``numpy`` arrays — this is the *boundary* type
handed across the wave-function protocol; device-resident walker state lives behind the C ABI, where the
same minimal-image convention is applied by ``min_image`` (``csrc/pqa_common.hpp``).
"""

import copy

import numpy as np


class RawDistance:
    """Open-boundary displacement conventions of ``pyqmc/configurations/distance.py:18-76``
    (``dist_i`` returns *b minus a*: new position minus the others)."""

    def dist_i(self, a, b):
        assert b.ndim <= 2
        return b[:, np.newaxis, :] - a

    def dist_matrix(self, configs):
        nconf, n = configs.shape[:2]
        if n < 2:
            return np.zeros((nconf, 0, 3)), []
        iu, ju = np.triu_indices(n, k=1)  # row-major (0,1),(0,2)... == distance.py:51-54
        return configs[:, iu, :] - configs[:, ju, :], list(zip(iu.tolist(), ju.tolist()))

    def pairwise(self, config1, config2):
        if config1.shape[1] == 0 or config2.shape[1] == 0:
            return np.zeros((config1.shape[0], 0, 3))
        return config2[:, np.newaxis, :] - config1[:, :, np.newaxis]


class _WalkerArrays:
    """What the two containers share: every per-walker array named in ``_arrays`` (``configs``, for periodic cells also the
    ``wrap`` counters) is moved, resampled, split, joined and reshaped in step.  Constructor arguments other than the arrays
    come from ``_like()``."""

    _arrays = ("configs",)

    def _like(self, **arrays):
        raise NotImplementedError

    def _get(self):
        return {k: getattr(self, k) for k in self._arrays}

    def _view(self, index):
        return self._like(**{k: v[index] for k, v in self._get().items()})

    def select_electrons(self, es):
        return self._view((slice(None), es))

    def mask(self, mask):
        return self._view(mask)

    def move(self, e, new, accept):
        """Electron e of the walkers flagged in ``accept`` takes the position (and counters) of the electron object ``new``."""
        for k, v in self._get().items():
            v[accept, e, :] = getattr(new, k)[accept, :]

    def resample(self, newinds):
        for k, v in self._get().items():
            setattr(self, k, v[newinds])

    def split(self, npartitions):
        parts = {k: np.array_split(v, npartitions) for k, v in self._get().items()}
        return [self._like(**{k: parts[k][i] for k in parts}) for i in range(npartitions)]

    def join(self, configslist, axis=0):
        for k in self._arrays:
            setattr(self, k, np.concatenate([getattr(c, k) for c in configslist], axis=axis))

    def copy(self):
        return copy.deepcopy(self)

    def reshape(self, shape):
        for k, v in self._get().items():
            setattr(self, k, v.reshape(shape))


class OpenElectron:
    """(nconf,3) or (nconf,naip,3) positions for one electron (``coord.py:21-28``)."""

    def __init__(self, epos, dist=None):
        self.configs = epos
        self.dist = dist if dist is not None else RawDistance()

    def mask(self, mask):
        return OpenElectron(self.configs[mask], dist=self.dist)


class OpenConfigs(_WalkerArrays):
    """(nconf,nelec,3) walker positions (interface of ``coord.py:31-112``)."""

    def __init__(self, configs, dist=None):
        self.configs = np.ascontiguousarray(configs, dtype=float)
        self.dist = dist if dist is not None else RawDistance()

    def _like(self, configs):
        return OpenConfigs(configs, dist=self.dist)

    def electron(self, e):
        return OpenElectron(self.configs[:, e], self.dist)

    def make_irreducible(self, e, vec, mask=True):
        return OpenElectron(vec, self.dist)

    def split(self, npartitions):  # (the pieces get distance objects of their own, as coord.py:64-66 builds them)
        return [OpenConfigs(c) for c in np.array_split(self.configs, npartitions)]


# ------------------------------------------------------------------------------------ periodic


def enforce_pbc(lattvecs, epos):
    """Fold positions into the cell spanned by the rows of ``lattvecs`` (``pbc/pbc.py:18-49``).

    Returns (positions in the cell, integer-valued float ``wrap`` such that
    ``epos == final + wrap @ lattvecs``).  Fractional coordinates are ``epos @ inv(lattvecs)`` and are
    split with ``divmod(., 1)`` exactly as the reference does, so the in-cell interval is [0, 1)."""
    frac = np.einsum("...ij,jk->...ik", epos, np.linalg.inv(lattvecs))
    wrap, rem = np.divmod(frac, 1)
    return np.dot(rem, lattvecs), wrap


def _is_diagonal(m, tol):
    return np.all(np.abs(m - np.diag(np.diagonal(m))) < tol)


class MinimalImageDistance(RawDistance):
    """Displacements reduced to the nearest periodic image (``distance.py:83-159``).

    ``kind`` is chosen like the reference (:95-107): "diagonal" (lattice vectors along x, y, z),
    "orthogonal" (mutually orthogonal rows) or "general" (argmin over the 27 neighbouring cells)."""

    def __init__(self, latvec):
        latvec = np.asarray(latvec, dtype=float)
        tol = 1e-10
        if _is_diagonal(latvec, tol):
            self.kind = "diagonal"
        elif _is_diagonal(latvec @ latvec.T, tol):
            self.kind = "orthogonal"
        else:
            self.kind = "general"
        self._latvec = latvec
        self._invvec = np.linalg.inv(latvec)
        grid = np.meshgrid(*[np.arange(3)] * 3)  # same cell order as distance.py:113-117
        self.point_list = np.stack([g.ravel() for g in grid], axis=0).T - 1
        self.shifts = self.point_list @ latvec

    def _minimal_dist(self, d):
        if self.kind == "diagonal":
            d = np.array(d, dtype=float)
            for i in range(3):
                L = self._latvec[i, i]
                d[..., i] = (d[..., i] + L / 2) % L - L / 2
            return d
        if self.kind == "orthogonal":
            frac = np.einsum("...ij,jk->...ik", d, self._invvec)
            frac = (frac + 0.5) % 1 - 0.5
            return np.einsum("...ij,jk->...ik", frac, self._latvec)
        cand = d[np.newaxis] + self.shifts.reshape((-1,) + (1,) * (d.ndim - 1) + (3,))
        best = np.argmin(np.sum(cand**2, axis=-1), axis=0)
        return np.take_along_axis(cand, best[np.newaxis, ..., np.newaxis], axis=0)[0]

    def dist_i(self, a, b):
        return self._minimal_dist(super().dist_i(a, b))

    def dist_matrix(self, configs):
        d, ij = super().dist_matrix(configs)
        return (self._minimal_dist(d) if len(ij) else d), ij

    def pairwise(self, config1, config2):
        return self._minimal_dist(super().pairwise(config1, config2))


class PeriodicElectron:
    """Positions of one electron inside the cell plus the integer ``wrap`` that brought them there
    (``coord.py:115-134``)."""

    def __init__(self, epos, lattice_vectors, dist, wrap=None):
        self.configs = epos
        self.lvecs = lattice_vectors
        self.wrap = wrap if wrap is not None else np.zeros_like(epos)
        self.dist = dist

    def mask(self, mask):
        return PeriodicElectron(self.configs[mask], self.lvecs, self.dist, wrap=self.wrap[mask])


class PeriodicConfigs(_WalkerArrays):
    """(nconf,nelec,3) walker positions folded into the simulation cell, with per-electron ``wrap`` counters
    (interface of ``coord.py:137-252``)."""

    _arrays = ("configs", "wrap")

    def __init__(self, configs, lattice_vectors, wrap=None, dist=None):
        self.lvecs = np.asarray(lattice_vectors, dtype=float)
        folded, crossed = enforce_pbc(self.lvecs, np.asarray(configs, dtype=float))
        self.configs = np.ascontiguousarray(folded)
        self.wrap = crossed if wrap is None else crossed + wrap
        self.dist = dist if dist is not None else MinimalImageDistance(self.lvecs)

    def _like(self, configs, wrap):
        return PeriodicConfigs(configs, self.lvecs, wrap=wrap, dist=self.dist)

    def electron(self, e):
        return PeriodicElectron(self.configs[:, e], self.lvecs, self.dist, wrap=self.wrap[:, e])

    def make_irreducible(self, e, vec, mask=None):
        """Electron object for proposed positions ``vec`` of electron ``e`` — (nconf,3), or (nconf,naip,3) for auxiliary points —
        folded into the cell where ``mask`` (default: everywhere) selects them; its counters are the electron's own plus the cells
        crossed by the fold (semantics of ``coord.py:164-178``)."""
        vec = np.asarray(vec)
        own = self.wrap[:, e, :]
        wrap = np.array(np.broadcast_to(own[:, np.newaxis, :] if vec.ndim == 3 else own, vec.shape))
        epos = vec.copy()
        where = np.ones(vec.shape[:-1], dtype=bool) if mask is None else mask
        epos[where], crossed = enforce_pbc(self.lvecs, vec[where])
        wrap[where] += crossed
        return PeriodicElectron(epos, self.lvecs, self.dist, wrap=wrap)
