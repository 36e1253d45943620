"""Oracle: Jastrow radial basis functions (test infrastructure).

Follows ``pyqmc/wf/func3d.py``: Pade polynomial ``PolyPadeFunction`` :25-109,
cusp function ``CutoffCuspFunction`` :112-210, cut-off evaluator
``CutoffFunc3dEvaluator`` :288-342 (zero for r >= rcut via ``r < rcut`` select),
default basis from ``pyqmc/wftools.py:64-96``.

A basis is a list of ("pade", beta) / ("cusp", gamma) entries sharing one rcut.
Each evaluator returns arrays with the basis index appended last, like the
reference (``np.moveaxis(out, 0, -1)``, :309).
"""

import numpy as np


def expand_beta(beta0, n):
    """wftools.py:64-73."""
    beta = np.zeros(n)
    if n == 0:
        return beta
    beta[0] = beta0
    b1 = np.log(beta0 + 1.00001)
    for i in range(1, n):
        beta[i] = np.exp(b1 + 1.6 * i) - 1
    return beta


def default_basis(ion_cusp=False, na=4, nb=3, rcut=7.5, gamma=24.0, beta_a=0.2, beta_b=0.5):
    """wftools.py:76-96 (molecular default rcut = 7.5)."""
    abasis = ([("cusp", float(gamma))] if ion_cusp else []) + [("pade", float(b)) for b in expand_beta(beta_a, na)]
    bbasis = [("cusp", float(gamma))] + [("pade", float(b)) for b in expand_beta(beta_b, nb)]
    return abasis, bbasis, float(rcut)


def _pade(r, beta, rcut):
    """value, (dU/dr)/r, laplacian — func3d.py:25-49."""
    z1 = r / rcut - 1.0
    z12 = z1 * z1
    p = (3.0 * z12 + 4.0 * z1) * z12 + 1.0
    obp = 1.0 / (1.0 + beta * p)
    val = (1.0 - p) * obp
    gfac = -(1.0 + beta) * 12.0 / rcut**2 * obp * obp * z12
    with np.errstate(divide="ignore", invalid="ignore"):
        lap = gfac * (5.0 + 2.0 / z1 - 24.0 * beta * (z1 + 1.0) ** 2 * z12 * obp)
    return val, gfac, lap


def _cusp(r, gamma, rcut):
    """value, (dU/dr)/r, laplacian — func3d.py:125-182."""
    y = r / rcut
    y1 = y - 1.0
    a = y1 * y1
    b = (a * y1 + 1.0) / 3.0
    ogb = 1.0 / (1.0 + gamma * b)
    val = (-b * ogb + 1.0 / (3.0 + gamma)) * rcut
    with np.errstate(divide="ignore", invalid="ignore"):
        c = ogb * ogb / r
        gfac = -a * c
        lap = -2.0 * c * ((y1 - a * a * gamma * ogb) * y + a)
    return val, gfac, lap


def evaluate(basis, rcut, d, r, want):
    """want in {"value","gradient_value","gradient_laplacian"}.

    Returns value (...,nbas)  /  (grad (...,nbas,3), value (...,nbas))  /
    (grad (...,nbas,3), lap (...,nbas)); all zero where r >= rcut (func3d.py:299-324)."""
    nb = len(basis)
    inside = r < rcut
    rs = np.where(inside, r, 0.5 * rcut)  # evaluate safely, mask afterwards
    val = np.zeros(r.shape + (nb,))
    gf = np.zeros(r.shape + (nb,))
    lap = np.zeros(r.shape + (nb,))
    for k, (kind, par) in enumerate(basis):
        v, g, l = (_pade if kind == "pade" else _cusp)(rs, par, rcut)
        val[..., k], gf[..., k], lap[..., k] = v, g, l
    m = inside[..., None]
    val = np.where(m, val, 0.0)
    if want == "value":
        return val
    grad = np.where(m[..., None], gf[..., None] * np.asarray(d)[..., None, :], 0.0)
    if want == "gradient_value":
        return grad, val
    return grad, np.where(m, lap, 0.0)
