"""Jastrow radial-function descriptors with the constructor signatures of
``pyqmc/wf/func3d.py`` (``PolyPadeFunction(beta, rcut)`` :52-66,
``CutoffCuspFunction(gamma, rcut)`` :112-123) and the default basis of
``pyqmc/wftools.py:64-96``.  They only carry parameters: evaluation happens in the
HIP kernels (``csrc/pqa_jastrow.hpp``)."""

import numpy as np


class PolyPadeFunction:
    kind = 0

    def __init__(self, beta, rcut):
        self.parameters = {"beta": float(beta), "rcut": float(rcut)}

    @property
    def param(self):
        return self.parameters["beta"]

    @property
    def rcut(self):
        return self.parameters["rcut"]


class CutoffCuspFunction:
    kind = 1

    def __init__(self, gamma, rcut):
        self.parameters = {"gamma": float(gamma), "rcut": float(rcut)}

    @property
    def param(self):
        return self.parameters["gamma"]

    @property
    def rcut(self):
        return self.parameters["rcut"]


def expand_beta_qwalk(beta0, n):
    """wftools.py:64-73."""
    beta = np.zeros(n)
    if n:
        beta[0] = beta0
        beta1 = np.log(beta0 + 1.00001)
        for i in range(1, n):
            beta[i] = np.exp(beta1 + 1.6 * i) - 1
    return beta


def default_jastrow_basis(mol, ion_cusp=False, na=4, nb=3, rcut=None, cusp_gamma=None, beta_a=0.2, beta_b=0.5):
    """wftools.py:76-96: rcut = 7.5 for molecules, min_i pi/|b_i| (half the smallest lattice-plane spacing)
    for a periodic cell."""
    cusp_gamma = 24 if cusp_gamma is None else cusp_gamma
    if rcut is None:
        rcut = float(np.amin(np.pi / np.linalg.norm(mol.reciprocal_vectors(), axis=1))) if hasattr(mol, "a") else 7.5
    abasis = [CutoffCuspFunction(cusp_gamma, rcut)] if ion_cusp else []
    abasis += [PolyPadeFunction(b, rcut) for b in expand_beta_qwalk(beta_a, na)]
    bbasis = [CutoffCuspFunction(cusp_gamma, rcut)] + [PolyPadeFunction(b, rcut) for b in expand_beta_qwalk(beta_b, nb)]
    return abasis, bbasis
