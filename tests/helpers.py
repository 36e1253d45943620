"""Shared builders for the parity tests: the same synthetic inputs the golden generator used."""

import ast
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from pyqmc_amd import systems  # noqa: E402
from pyqmc_amd.configs import OpenConfigs  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def jastrow_params(mol, seed=11, na=4, nb=4):
    """Same draws as make_golden.make_wf."""
    rng = np.random.default_rng(seed)
    acoeff = 0.05 * rng.standard_normal((mol.natm, na, 2))
    bcoeff = 0.05 * rng.standard_normal((nb, 3))
    bcoeff[0] = [-0.25, -0.5, -0.25]
    return acoeff, bcoeff


def oracle_wf(mol, mf, determinants=None, seed=11):
    from oracle import jastrow_basis, wf as owf

    sl = owf.Slater(mol, mf.mo_coeff, determinants)
    ab, bb, rcut = jastrow_basis.default_basis(ion_cusp=False)
    ja = owf.JastrowSpin(mol, ab, bb, rcut)
    ja.parameters["acoeff"], ja.parameters["bcoeff"] = jastrow_params(mol, seed)
    return owf.MultiplyWF(sl, ja)


def gpu_wf(mol, mf, determinants=None, seed=11):
    import pyqmc_amd as pa

    wf = pa.generate_wf(mol, mf, determinants=determinants)
    a, b = jastrow_params(mol, seed)
    wf.parameters["wf2acoeff"] = a
    wf.parameters["wf2bcoeff"] = b
    return wf


def case(name):
    """(mol, mf, determinants, golden-dict) for the protocol fixtures."""
    g = golden(name)
    if name == "g5_protocol_h2o":
        mol = systems.water()
        return mol, systems.random_mf(mol), None, g
    if name == "g8_protocol_h2o_multidet":
        mol = systems.water()
        return mol, systems.random_mf(mol, nvirt=6), ast.literal_eval(str(g["det_json"])), g
    if name == "g5_protocol_cluster":
        mol = systems.water_cluster()
        return mol, systems.random_mf(mol), None, g
    raise KeyError(name)


def relerr(a, b):
    a, b = np.asarray(a), np.asarray(b)
    scale = max(np.max(np.abs(b)), 1e-300) if b.size else 1.0
    return float(np.max(np.abs(a - b)) / scale) if b.size else 0.0


def relerr_elem(a, b, floor=1e-6):
    """Element-wise relative error max |a - b| / (|b| + floor * max|b|): entries larger than ``floor`` times the largest one
    are compared RELATIVELY (``relerr`` above measures everything against the global maximum, so the small entries of, say,
    an inverse with |max| ~ 200 are only checked absolutely); the floor keeps exact zeros and cancellation noise finite."""
    a, b = np.asarray(a), np.asarray(b)
    if not b.size:
        return 0.0
    scale = np.abs(b) + floor * max(float(np.max(np.abs(b))), 1e-300)
    return float(np.max(np.abs(a - b) / scale))


def g5_tolerance(key, loose=1.0):
    """SURVEY.md section 8(c) G5 tolerances: ratios / values / logs 1e-11, gradients 1e-10, Laplacians 1e-9 (element-wise
    relative, helpers.relerr_elem); ``loose`` scales them for ill-conditioned fixtures."""
    if "lap" in key:
        return 1e-9 * loose
    if "grad" in key:
        return 1e-10 * loose
    return 1e-11 * loose


def run_protocol(wf, g, names=("slater", "jastrow", "wf"), relerr=relerr):
    """Replay ``make_golden.protocol_dump`` on ``wf`` (oracle or HIP) and return
    {quantity: relative error vs golden}."""
    configs = OpenConfigs(g["configs"].copy())
    factors = {"slater": wf.wf_factors[0], "jastrow": wf.wf_factors[1], "wf": wf}
    factors = {k: v for k, v in factors.items() if k in names}
    err = {}
    for nm, w in factors.items():
        s, l = w.recompute(configs)
        err[f"{nm}_recompute_log"] = relerr(l, g[f"{nm}_recompute_log"])
        err[f"{nm}_recompute_sign"] = relerr(s, g[f"{nm}_recompute_sign"])
    for e in g["electrons"]:
        e = int(e)
        ep = configs.make_irreducible(e, g[f"e{e}_newpos"])
        ea = configs.make_irreducible(e, g[f"e{e}_aux"])
        mask, accept = g[f"e{e}_mask"], g[f"e{e}_accept"]
        for nm, w in factors.items():
            p = f"e{e}_{nm}_"
            gr, v, _ = w.gradient_value(e, ep)
            err[p + "gv_grad"], err[p + "gv_val"] = relerr(gr, g[p + "gv_grad"]), relerr(v, g[p + "gv_val"])
            err[p + "grad"] = relerr(w.gradient(e, ep), g[p + "grad"])
            gr, l = w.gradient_laplacian(e, ep)
            err[p + "gl_grad"], err[p + "gl_lap"] = relerr(gr, g[p + "gl_grad"]), relerr(l, g[p + "gl_lap"])
            gr, l = w.gradient_laplacian(e, configs.electron(e))
            err[p + "gl0_grad"], err[p + "gl0_lap"] = relerr(gr, g[p + "gl0_grad"]), relerr(l, g[p + "gl0_lap"])
            err[p + "testvalue"] = relerr(w.testvalue(e, ep)[0], g[p + "testvalue"])
            err[p + "testvalue_mask"] = relerr(w.testvalue(e, ep, mask)[0], g[p + "testvalue_mask"])
            err[p + "testvalue_aux"] = relerr(w.testvalue(e, ea, mask)[0], g[p + "testvalue_aux"])
        if "wf" in factors:
            _, _, saved = wf.gradient_value(e, ep)
            configs.move(e, ep, accept)
            wf.updateinternals(e, ep, configs, mask=accept, saved_values=saved)
        else:
            configs.move(e, ep, accept)
            for w in factors.values():
                w.updateinternals(e, ep, configs, mask=accept)
        for nm, w in factors.items():
            s, l = w.value()
            err[f"e{e}_{nm}_post_log"] = relerr(l, g[f"e{e}_{nm}_post_log"])
            err[f"e{e}_{nm}_post_sign"] = relerr(s, g[f"e{e}_{nm}_post_sign"])
    return err


class ReplayTape:
    """Replays the reference's recorded random draws (tests/golden/g12_dmc.npz) in call order."""

    def __init__(self, g):
        self._n, self._rs, self._rv = iter(g["normal"]), iter(g["rand_scalar"]), iter(g["rand_vector"])
        self._rot, self._rnd = iter(g["rot"]), iter(g["random"])

    def normal(self, W):
        return next(self._n)

    def rand(self, W):
        return next(self._rv)

    def rand1(self):
        return float(next(self._rs))

    def rot(self):
        return next(self._rot)

    def random(self, W):
        return next(self._rnd)


def ccoeff_params(mol, na=4, nb=4, seed=12):
    return 0.1 * np.random.default_rng(seed).standard_normal((mol.natm, na, na, nb, 3))


def oracle_wf3(mol, mf, determinants=None):
    """Slater x two-body x three-body Jastrow (config C4 shape), parameters as in make_golden.g_jastrow3."""
    from oracle import jastrow_basis, wf as owf

    base = oracle_wf(mol, mf, determinants)
    ab, bb, rcut = jastrow_basis.default_basis(ion_cusp=False)
    j3 = owf.ThreeBodyJastrow(mol, ab, bb, rcut)
    j3.parameters["ccoeff"] = ccoeff_params(mol)
    return owf.MultiplyWF(base.wf_factors[0], base.wf_factors[1], j3)


def gpu_wf3(mol, mf, determinants=None):
    import pyqmc_amd as pa

    wf = pa.generate_wf(mol, mf, determinants=determinants, jastrow3=True)
    a, b = jastrow_params(mol)
    wf.parameters["wf2acoeff"] = a
    wf.parameters["wf2bcoeff"] = b
    wf.parameters["wf3ccoeff"] = ccoeff_params(mol)
    return wf


def run_protocol3(wf, g):
    """Replay make_golden.g_jastrow3's protocol section on the three-body factor and on the product."""
    configs = OpenConfigs(g["configs"].copy())
    names = {"j3": wf.wf_factors[2], "wf": wf}
    err = {}
    for nm, w in names.items():
        err[f"{nm}_recompute_log"] = relerr(w.recompute(configs)[1], g[f"{nm}_recompute_log"])
    for e in g["electrons"]:
        e = int(e)
        ep = configs.make_irreducible(e, g[f"e{e}_newpos"])
        ea = configs.make_irreducible(e, g[f"e{e}_aux"])
        mask, accept = g[f"e{e}_mask"], g[f"e{e}_accept"]
        for nm, w in names.items():
            p = f"e{e}_{nm}_"
            gr, v, _ = w.gradient_value(e, ep)
            err[p + "gv_grad"], err[p + "gv_val"] = relerr(gr, g[p + "gv_grad"]), relerr(v, g[p + "gv_val"])
            err[p + "grad"] = relerr(w.gradient(e, ep), g[p + "grad"])
            gr, l = w.gradient_laplacian(e, ep)
            err[p + "gl_grad"], err[p + "gl_lap"] = relerr(gr, g[p + "gl_grad"]), relerr(l, g[p + "gl_lap"])
            err[p + "testvalue"] = relerr(w.testvalue(e, ep)[0], g[p + "testvalue"])
            err[p + "testvalue_aux"] = relerr(w.testvalue(e, ea, mask)[0], g[p + "testvalue_aux"])
        configs.move(e, ep, accept)
        wf.updateinternals(e, ep, configs, mask=accept)
        for nm, w in names.items():
            err[f"e{e}_{nm}_post_log"] = relerr(w.value()[1], g[f"e{e}_{nm}_post_log"])
    for nm, w in names.items():
        err[f"{nm}_final_recompute_log"] = relerr(w.recompute(configs)[1], g[f"{nm}_final_recompute_log"])
    return err


def run_protocol_pbc(names, g, prefix, cell, update_first=False):
    """Replay make_golden.pbc_protocol on wave-function factors (oracle or HIP) living on PeriodicConfigs.
    ``names``: {golden name: factor}; the last one receives updateinternals.  Also checks the container itself
    (folded positions and wrap counters of make_irreducible / move) against the reference's."""
    from pyqmc_amd.configs import PeriodicConfigs

    configs = PeriodicConfigs(g[prefix + "configs"].copy(), cell.lattice_vectors(), wrap=g[prefix + "wrap"].copy())
    err = {}
    for nm, w in names.items():
        s, l = w.recompute(configs)
        err[f"{nm}_recompute_log"] = relerr(l, g[f"{prefix}{nm}_recompute_log"])
        err[f"{nm}_recompute_sign"] = relerr(s, g[f"{prefix}{nm}_recompute_sign"])
    top = list(names.values())[-1]
    for e in g[prefix + "electrons"]:
        e = int(e)
        q = f"{prefix}e{e}_"
        mask, accept = g[q + "mask"], g[q + "accept"]
        ep = configs.make_irreducible(e, g[q + "newpos"])
        ea = configs.make_irreducible(e, g[q + "aux"], mask)
        err[f"e{e}_container"] = max(relerr(ep.configs, g[q + "ep_configs"]), relerr(ep.wrap, g[q + "ep_wrap"]),
                                     relerr(ea.configs, g[q + "ea_configs"]), relerr(ea.wrap, g[q + "ea_wrap"]))
        for nm, w in names.items():
            p = f"{q}{nm}_"
            o = f"e{e}_{nm}_"
            gr, v, _ = w.gradient_value(e, ep)
            err[o + "gv_grad"], err[o + "gv_val"] = relerr(gr, g[p + "gv_grad"]), relerr(v, g[p + "gv_val"])
            err[o + "grad"] = relerr(w.gradient(e, ep), g[p + "grad"])
            gr, l = w.gradient_laplacian(e, ep)
            err[o + "gl_grad"], err[o + "gl_lap"] = relerr(gr, g[p + "gl_grad"]), relerr(l, g[p + "gl_lap"])
            gr, l = w.gradient_laplacian(e, configs.electron(e))
            err[o + "gl0_grad"], err[o + "gl0_lap"] = relerr(gr, g[p + "gl0_grad"]), relerr(l, g[p + "gl0_lap"])
            err[o + "testvalue"] = relerr(w.testvalue(e, ep)[0], g[p + "testvalue"])
            err[o + "testvalue_mask"] = relerr(w.testvalue(e, ep, mask)[0], g[p + "testvalue_mask"])
            err[o + "testvalue_aux"] = relerr(w.testvalue(e, ea, mask)[0], g[p + "testvalue_aux"])
        if update_first:
            top.updateinternals(e, ep, configs, mask=accept)
            configs.move(e, ep, accept)
        else:
            configs.move(e, ep, accept)
            top.updateinternals(e, ep, configs, mask=accept)
        for nm, w in names.items():
            s, l = w.value()
            err[f"e{e}_{nm}_post_log"] = relerr(l, g[f"{q}{nm}_post_log"])
    err["final_container"] = max(relerr(configs.configs, g[prefix + "final_configs"]), relerr(configs.wrap, g[prefix + "final_wrap"]))
    for nm, w in names.items():
        err[f"{nm}_final_recompute_log"] = relerr(w.recompute(configs)[1], g[f"{prefix}{nm}_final_recompute_log"])
    return err


PBC_JASTROW_CASES = {"cubic": (systems.diamond_cubic, {}), "prim": (systems.diamond_primitive, {"rcut": 4.0})}


def pbc_jastrow_coeffs(cell, nb=4, na=4):
    rng = np.random.default_rng(17)
    a = 0.05 * rng.standard_normal((cell.natm, na, 2))
    b = 0.05 * rng.standard_normal((nb, 3))
    b[0] = [-0.25, -0.5, -0.25]
    return a, b


PBC_SLATER_CASES = {
    "gamma": np.eye(3),
    "fcc2cubic": np.array([[-1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 1.0, -1.0]]),
    "k222": 2.0 * np.eye(3),
}


def pbc_slater_case(tag):
    """(supercell, k-point mean field) exactly as make_golden.g_pbc_slater built them."""
    from pyqmc_amd import pbc

    sup = pbc.get_supercell(systems.diamond_primitive(), PBC_SLATER_CASES[tag])
    return sup, pbc.random_kmf(sup)


def unfold_ao(sup, kpts, ao_super):
    """(…, nao_super) Gamma-point AOs of the supercell -> (nk, …, nao_prim) Bloch sums sum_c e^{ik.T_c} AO[(a,c,mu)]."""
    from pyqmc_amd import pbc

    prim = sup.original_cell
    copies = pbc.get_supercell_copies(prim.lattice_vectors(), sup.S)
    phase = np.exp(1j * copies @ np.asarray(kpts).T).real  # (ncopy, nk)
    nao_atom = [sum(2 * sh[0] + 1 for sh in prim._basis[n]) for n in prim._names]
    out = np.zeros((len(kpts),) + ao_super.shape[:-1] + (sum(nao_atom),))
    row = col = 0
    for na in nao_atom:
        for c in range(len(copies)):
            for k in range(len(kpts)):
                out[k][..., col : col + na] += phase[c, k] * ao_super[..., row : row + na]
            row += na
        col += na
    return out


def oracle_pbc_wf(tag, Ls=None):
    """Oracle Slater x Jastrow for a PBC_SLATER_CASES entry, parameters as make_golden.ref_pbc_wf."""
    from oracle import jastrow_basis, wf as owf
    from pyqmc_amd import pbc

    sup, mf = pbc_slater_case(tag)
    if Ls is None:
        Ls = pbc.lattice_points_within(sup.original_cell.lattice_vectors(), 30.0)
    sl = owf.Slater.periodic(sup, mf.kpts, mf.mo_coeff, Ls)
    rcut = float(np.amin(np.pi / np.linalg.norm(sup.reciprocal_vectors(), axis=1)))
    ab, bb, rcut = jastrow_basis.default_basis(ion_cusp=False, rcut=rcut)
    ja = owf.JastrowSpin(sup, ab, bb, rcut)
    ja.parameters["acoeff"], ja.parameters["bcoeff"] = pbc_jastrow_coeffs(sup)
    return sup, owf.MultiplyWF(sl, ja)


def gpu_pbc_wf(tag, **kw):
    import pyqmc_amd as pa

    sup, mf = pbc_slater_case(tag)
    wf = pa.generate_wf(sup, mf, **kw)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    return sup, wf


def madelung_cases():
    """The reference's known-answer Ewald systems (tests/unit/test_ewald.py:37-66, :141-184), restated as data:
    [(name, supercell, configs (1,N,3), expected total Coulomb energy of the cell)].  Point charges: +1 / +2 ions at
    the lattice sites, electrons at the anion sites; nearest-neighbour distance 1."""
    from pyqmc_amd import pbc

    out = []
    L = 2.0
    nacl = systems.Cell(["H"], [(0.0, 0.0, 0.0)], (np.ones((3, 3)) - np.eye(3)) * L / 2, nelec=(1, 0), ecp={})
    out.append(("nacl_prim", pbc.get_supercell(nacl, np.eye(3)), np.ones((1, 1, 3)) * L / 2, -1.74756))
    cfg = np.ones((1, 4, 3)) * L / 2
    cfg[:, 1:, :] = np.eye(3) * L / 2
    out.append(("nacl_conv", pbc.get_supercell(nacl, np.ones((3, 3)) - 2 * np.eye(3)), cfg, -4 * 1.74756))
    L = 4 / np.sqrt(3)
    caf2 = systems.Cell(["He"], [(0.0, 0.0, 0.0)], (np.ones((3, 3)) - np.eye(3)) * L / 2, nelec=(1, 1), ecp={})
    cfg = np.ones((1, 2, 3)) * L / 4
    cfg[0, 1, 1] *= -1
    out.append(("caf2_prim", pbc.get_supercell(caf2, np.eye(3)), cfg, -5.03879))
    cube = np.stack(np.meshgrid(*[[0, 1]] * 3, indexing="ij"), axis=-1).reshape((-1, 3))
    out.append(("caf2_conv", pbc.get_supercell(caf2, np.ones((3, 3)) - 2 * np.eye(3)), np.reshape((cube + 0.5) * L / 2, (1, 8, 3)), -4 * 5.03879))
    return out


def pbc_complex_case():
    """3x1x1 diamond supercell with complex Bloch coefficients at k = 0, 1/3, 2/3 b1 (make_golden.ref_pbc_wf_complex)."""
    from pyqmc_amd import pbc

    sup = pbc.get_supercell(systems.diamond_primitive(), np.diag([3.0, 1.0, 1.0]))
    return sup, pbc.random_kmf(sup, complex_coeff=True)


TWIST_CASES = {"prim": (np.eye(3), (0.25, 0.1, -0.3)), "s211": (np.diag([2.0, 1.0, 1.0]), (0.2, -0.15, 0.4)),
               "s222": (2.0 * np.eye(3), (0.2, -0.15, 0.4))}  # s222: 32 electrons per spin (no golden; device paths against each other)


def twist_case(tag):
    """Twisted diamond cells of make_golden.ref_twisted_wf: (supercell, k-point mean field with complex coefficients)."""
    from pyqmc_amd import pbc

    S, twist = TWIST_CASES[tag]
    sup = pbc.get_supercell(systems.diamond_primitive(), S)
    return sup, pbc.random_kmf(sup, complex_coeff=True, twist=twist)
