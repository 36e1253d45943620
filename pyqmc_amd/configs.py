"""Walker containers with the interface of ``pyqmc/configurations/coord.py:21-112``.

Only the open-boundary container is provided in this round (the PBC container,
``coord.py:137-252``, belongs to the diamond configs C3/C5).  The arrays are
ordinary host ``numpy`` arrays — this is the *boundary* type handed across the
wave-function protocol; device-resident walker state lives behind the C ABI.
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


class OpenElectron:
    """(nconf,3) or (nconf,naip,3) positions for one electron (``coord.py:21-28``)."""

    def __init__(self, epos, dist=None):
        self.configs = epos
        self.dist = dist if dist is not None else RawDistance()

    def mask(self, mask):
        return OpenElectron(self.configs[mask], dist=self.dist)


class OpenConfigs:
    """(nconf,nelec,3) walker positions (``coord.py:31-112``)."""

    def __init__(self, configs, dist=None):
        self.configs = np.ascontiguousarray(configs, dtype=float)
        self.dist = dist if dist is not None else RawDistance()

    def electron(self, e):
        return OpenElectron(self.configs[:, e], self.dist)

    def select_electrons(self, es):
        return OpenConfigs(self.configs[:, es], self.dist)

    def mask(self, mask):
        return OpenConfigs(self.configs[mask], dist=self.dist)

    def make_irreducible(self, e, vec, mask=True):
        return OpenElectron(vec, self.dist)

    def move(self, e, new, accept):
        self.configs[accept, e, :] = new.configs[accept, :]

    def resample(self, newinds):
        self.configs = self.configs[newinds]

    def split(self, npartitions):
        return [OpenConfigs(c) for c in np.array_split(self.configs, npartitions)]

    def join(self, configslist, axis=0):
        self.configs = np.concatenate([c.configs for c in configslist], axis=axis)

    def copy(self):
        return copy.deepcopy(self)

    def reshape(self, shape):
        self.configs = self.configs.reshape(shape)
