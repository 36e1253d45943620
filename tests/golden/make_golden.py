"""Generate golden vectors from the REAL reference (run in the build container only).

    python tests/golden/make_golden.py

Imports WagnerGroup/pyqmc from /root/reference with pyscf/h5py mocked and numba
replaced by an identity decorator (so ``pyqmc/wf/numba/gto.py`` runs as plain IEEE
fp64 Python), feeds it the duck-typed systems of ``pyqmc_amd.systems`` and stores
inputs + outputs as small ``.npz`` files next to this script.  Only data is stored;
no reference source travels.  All randomness the reference draws (proposal
Gaussians, Metropolis uniforms, ECP mask uniforms, ECP quadrature rotations) is
routed through recorded tapes so the oracle and the HIP path can replay it.
"""

import os
import sys
import types
from unittest import mock

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _identity(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
        return a[0]
    return lambda f: f


nb = types.ModuleType("numba")
nb.njit = nb.jit = _identity
sys.modules["numba"] = nb
for m in ["pyscf", "pyscf.pbc", "pyscf.pbc.gto", "pyscf.pbc.gto.eval_gto", "pyscf.pbc.gto.cell", "pyscf.mcscf",
          "pyscf.fci", "pyscf.hci", "pyscf.gto", "pyscf.lib", "pyscf.scf", "pyscf.pbc.scf", "pyscf.pbc.scf.addons", "h5py"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import scipy.spatial.transform  # noqa: E402
import pyqmc.api as pyq  # noqa: E402
import pyqmc.wf.numba.gto as refgto  # noqa: E402
import pyqmc.wf.func3d as func3d  # noqa: E402
import pyqmc.observables.eval_ecp as eval_ecp  # noqa: E402
import pyqmc.observables.energy as refenergy  # noqa: E402
from pyqmc.method.mc import vmc_worker  # noqa: E402
from pyqmc.configurations.coord import OpenConfigs  # noqa: E402
from pyqmc.wf.slater import sherman_morrison_ms  # noqa: E402

from pyqmc_amd import systems  # noqa: E402


class Tapes:
    """Route the reference's global-RNG draws through seeded generators and record them."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.log = {"normal": [], "rand": [], "random": [], "rot": []}

    def normal(self, loc=0.0, scale=1.0, size=None):
        z = self.rng.standard_normal(size)
        self.log["normal"].append(z)
        return loc + scale * z

    def rand(self, *shape):
        u = self.rng.random(shape)
        self.log["rand"].append(u)
        return u

    def random(self, size=None):
        u = self.rng.random(size)
        self.log["random"].append(u)
        return u

    def rot(self):
        q = self.rng.standard_normal(4)
        q /= np.linalg.norm(q)
        w, x, y, z = q
        R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                      [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                      [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        self.log["rot"].append(R)
        return types.SimpleNamespace(as_matrix=lambda R=R: R)

    def __enter__(self):
        # eval_ecp.get_rot calls ``scipy.spatial.transform.Rotation.random()`` through the module
        # global ``scipy`` (eval_ecp.py:17,263); Rotation is an immutable type, so shadow the global.
        self._saved = (np.random.normal, np.random.rand, np.random.random, eval_ecp.scipy)
        np.random.normal, np.random.rand, np.random.random = self.normal, self.rand, self.random
        ns = types.SimpleNamespace
        eval_ecp.scipy = ns(spatial=ns(transform=ns(Rotation=ns(random=self.rot))))
        return self

    def __exit__(self, *a):
        np.random.normal, np.random.rand, np.random.random, eval_ecp.scipy = self._saved


def make_wf(mol, mf, determinants=None, seed=11):
    kws = dict(evaluate_orbitals_with="numba")
    if determinants is not None:
        kws["determinants"] = determinants
    wf, _ = pyq.generate_wf(mol, mf, slater_kws=kws)
    rng = np.random.default_rng(seed)
    jas = wf.wf_factors[1]
    jas.parameters["acoeff"] = 0.05 * rng.standard_normal(jas.parameters["acoeff"].shape)
    b = 0.05 * rng.standard_normal(jas.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    jas.parameters["bcoeff"] = b
    return wf


def walkers(mol, W, seed):
    return OpenConfigs(systems.initial_guess(mol, W, rng=np.random.default_rng(seed)).configs.copy())


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ------------------------------------------------------------------ G1 Sherman-Morrison
def g_sherman_morrison():
    rng = np.random.default_rng(1)
    out = {}
    for tag, (n, W, D, e) in {"small": (10, 4, 8, 2), "n32": (32, 3, 1, 17)}.items():
        mats = rng.standard_normal((W, D, n, n)) + 2.0 * np.eye(n)
        inv = np.linalg.inv(mats)
        vec = rng.standard_normal((W, D, n))
        ratio, invnew = sherman_morrison_ms(e, inv, vec)
        out.update({f"{tag}_inv": inv, f"{tag}_vec": vec, f"{tag}_e": e, f"{tag}_ratio": ratio, f"{tag}_invnew": invnew})
    save("g1_sherman_morrison", **out)


# ------------------------------------------------------------------ G2 AO
def g_ao():
    rng = np.random.default_rng(2)
    out = {}
    for tag, mol in {"h2o": systems.water(), "c2": systems.carbon_dimer()}.items():
        ev = refgto.AtomicOrbitalEvaluator(mol)
        pts = np.concatenate([rng.standard_normal((24, 3)) * 1.5 + mol.atom_coords()[rng.integers(mol.natm, size=24)],
                              np.linspace(0, 7, 12)[:, None] * np.ones(3)[None]])
        out[f"{tag}_pts"] = pts
        out[f"{tag}_val"] = ev.eval_gto("GTOval_sph", pts)
        out[f"{tag}_deriv1"] = ev.eval_gto("GTOval_sph_deriv1", pts)
        out[f"{tag}_deriv2"] = ev.eval_gto("GTOval_sph_deriv2", pts)
    save("g2_ao", **out)


def g_ao_general():
    """G38: a generally contracted all-electron basis (systems.water_general: two s contractions over eight primitives, a p contraction
    sharing its exponents with an uncontracted function) — the reference's in-repo evaluator (numba/gto.py:435-470) takes one coefficient
    column per shell, so it is fed the single-column shells tables.split_general_contractions makes of the PySCF ``_basis`` entries (which
    is what ``mol._basis`` of a systems.Mol holds); the oracle and the device do their own split of the generally contracted tables."""
    rng = np.random.default_rng(38)
    mol = systems.water_general()
    ev = refgto.AtomicOrbitalEvaluator(mol)
    pts = np.concatenate([rng.standard_normal((30, 3)) * 1.2 + mol.atom_coords()[rng.integers(mol.natm, size=30)],
                          np.linspace(0.0, 5, 9)[:, None] * np.array([0.5, 0.4, -0.7])[None]])
    save("g38_ao_general", pts=pts, val=ev.eval_gto("GTOval_sph", pts), deriv1=ev.eval_gto("GTOval_sph_deriv1", pts),
         deriv2=ev.eval_gto("GTOval_sph_deriv2", pts))


def g_ao_high_l():
    """f, g and h shells (l = 3..5): the reference's AO evaluator (numba/gto.py:89-254 with the sphericart harmonics
    numba/spherical_harmonics.py:636-739) on a carbon dimer with added high-l shells."""
    rng = np.random.default_rng(12)
    mol = systems.carbon_dimer_high_l()
    ev = refgto.AtomicOrbitalEvaluator(mol)
    pts = np.concatenate([rng.standard_normal((40, 3)) * 1.2 + mol.atom_coords()[rng.integers(mol.natm, size=40)],
                          np.linspace(0.1, 5, 8)[:, None] * np.array([0.3, -0.5, 0.8])[None]])
    save("g26_ao_high_l", pts=pts, val=ev.eval_gto("GTOval_sph", pts), deriv1=ev.eval_gto("GTOval_sph_deriv1", pts),
         deriv2=ev.eval_gto("GTOval_sph_deriv2", pts), max_l=np.array(ev.max_l))


# ------------------------------------------------------------------ G4 func3d
def g_func3d():
    r = np.concatenate([np.linspace(0.05, 2.2, 40), [1.5, 1.5 - 1e-9, 1.5 + 1e-9, 7.4999, 7.5, 9.0]])
    rvec = np.stack([0.6 * r, 0.0 * r, 0.8 * r], axis=-1)
    out = {"r": r, "rvec": rvec}
    funcs = {"pade_2.0_1.5": func3d.PolyPadeFunction(2.0, 1.5), "cusp_2.0_1.5": func3d.CutoffCuspFunction(2.0, 1.5),
             "pade_0.2_7.5": func3d.PolyPadeFunction(0.2, 7.5), "cusp_24_7.5": func3d.CutoffCuspFunction(24, 7.5)}
    for k, f in funcs.items():
        out[k + "_value"] = f.value(rvec, r)
        g, v = f.gradient_value(rvec, r)
        out[k + "_grad"], out[k + "_gv_value"] = g, v
        g, l = f.gradient_laplacian(rvec, r)
        out[k + "_lap"] = l
    ev = func3d.CutoffFunc3dEvaluator([func3d.CutoffCuspFunction(24, 1.5), func3d.PolyPadeFunction(0.5, 1.5)], 1.5)
    out["eval_value"] = ev.value(rvec, r)
    out["eval_grad"], out["eval_gv_value"] = ev.gradient_value(rvec, r)
    out["eval_gl_grad"], out["eval_lap"] = ev.gradient_laplacian(rvec, r)
    save("g4_func3d", **out)


# ------------------------------------------------------------------ G5-G8 wave-function protocol
def protocol_dump(prefix, mol, mf, wf, W, seed, electrons, out, naip=6):
    """The update/testvalue/recompute triangle of tests/unit/test_wf_derivatives.py applied to
    each factor and to the product; everything needed to replay is stored."""
    rng = np.random.default_rng(seed)
    configs = walkers(mol, W, seed)
    out[prefix + "configs"] = configs.configs.copy()
    sl, ja = wf.wf_factors
    for k in ("det_coeff", "mo_coeff_alpha", "mo_coeff_beta"):
        out[prefix + k] = np.asarray(sl.parameters[k])
    out[prefix + "det_occup_up"] = np.asarray(sl._det_occup[0])
    out[prefix + "det_occup_dn"] = np.asarray(sl._det_occup[1])
    out[prefix + "det_map"] = np.asarray(sl._det_map)
    out[prefix + "acoeff"], out[prefix + "bcoeff"] = ja.parameters["acoeff"], ja.parameters["bcoeff"]
    names = {"slater": sl, "jastrow": ja, "wf": wf}
    for nm, w in names.items():
        s, l = w.recompute(configs)
        out[f"{prefix}{nm}_recompute_sign"], out[f"{prefix}{nm}_recompute_log"] = s, l
    for s in (0, 1):
        out[f"{prefix}slater_dets{s}"] = sl._dets[s].copy()
        out[f"{prefix}slater_inverse{s}"] = sl._inverse[s].copy()
    out[prefix + "jastrow_avalues"], out[prefix + "jastrow_bvalues"] = ja._avalues.copy(), ja._bvalues.copy()
    out[prefix + "electrons"] = np.asarray(electrons)
    for e in electrons:
        newpos = configs.configs[:, e, :] + 0.3 * rng.standard_normal((W, 3))
        aux = configs.configs[:, e, None, :] + 0.4 * rng.standard_normal((W, naip, 3))
        mask = rng.random(W) > 0.35
        mask[0] = True
        accept = rng.random(W) > 0.4
        out[f"{prefix}e{e}_newpos"], out[f"{prefix}e{e}_aux"] = newpos, aux
        out[f"{prefix}e{e}_mask"], out[f"{prefix}e{e}_accept"] = mask, accept
        ep = configs.make_irreducible(e, newpos)
        ea = configs.make_irreducible(e, aux)
        for nm, w in names.items():
            p = f"{prefix}e{e}_{nm}_"
            g, v, saved = w.gradient_value(e, ep)
            out[p + "gv_grad"], out[p + "gv_val"] = g, v
            out[p + "grad"] = w.gradient(e, ep)
            g, l = w.gradient_laplacian(e, ep)
            out[p + "gl_grad"], out[p + "gl_lap"] = g, l
            g, l = w.gradient_laplacian(e, configs.electron(e))
            out[p + "gl0_grad"], out[p + "gl0_lap"] = g, l
            out[p + "testvalue"] = w.testvalue(e, ep)[0]
            out[p + "testvalue_mask"] = w.testvalue(e, ep, mask)[0]
            out[p + "testvalue_aux"] = w.testvalue(e, ea, mask)[0]
        # masked update with the saved values of the product's gradient_value, like mc.py:124-136
        _, _, saved = wf.gradient_value(e, ep)
        configs.move(e, ep, accept)
        wf.updateinternals(e, ep, configs, mask=accept, saved_values=saved)
        for nm, w in names.items():
            s, l = w.value()
            out[f"{prefix}e{e}_{nm}_post_sign"], out[f"{prefix}e{e}_{nm}_post_log"] = s, l
        spin = int(e >= mol.nelec[0])
        out[f"{prefix}e{e}_post_inverse"] = sl._inverse[spin].copy()
        out[f"{prefix}e{e}_post_avalues"], out[f"{prefix}e{e}_post_bvalues"] = ja._avalues.copy(), ja._bvalues.copy()
    out[prefix + "final_configs"] = configs.configs.copy()
    for nm, w in names.items():
        s, l = w.recompute(configs)
        out[f"{prefix}{nm}_final_recompute_sign"], out[f"{prefix}{nm}_final_recompute_log"] = s, l


def g_protocol():
    out = {}
    mol = systems.water()
    mf = systems.random_mf(mol)
    protocol_dump("", mol, mf, make_wf(mol, mf), W=6, seed=5, electrons=[0, 2, 5, 7], out=out)
    save("g5_protocol_h2o", **out)

    out = {}
    mf = systems.random_mf(mol, nvirt=6)
    dets = systems.random_determinants(mol, mf, 12)
    out["det_json"] = np.asarray(repr(dets))
    protocol_dump("", mol, mf, make_wf(mol, mf, determinants=dets), W=5, seed=8, electrons=[1, 6], out=out)
    save("g8_protocol_h2o_multidet", **out)

    out = {}
    mol = systems.water_cluster()
    mf = systems.random_mf(mol)
    protocol_dump("", mol, mf, make_wf(mol, mf), W=3, seed=9, electrons=[0, 31, 32, 63], out=out)
    save("g5_protocol_cluster", **out)


# ------------------------------------------------------------------ G9/G10 ECP + energy
def g_energy():
    out = {}
    for tag, mol, W in (("h2o", systems.water(), 8), ("cluster", systems.water_cluster(), 2)):
        mf = systems.random_mf(mol)
        wf = make_wf(mol, mf)
        configs = walkers(mol, W, 21)
        # pull a few electrons close to an oxygen so the stochastic ECP mask has both outcomes
        rng = np.random.default_rng(3)
        configs.configs[:, :3, :] = mol.atom_coords()[0] + 0.35 * rng.standard_normal((W, 3, 3))
        out[tag + "_configs"] = configs.configs.copy()
        wf.recompute(configs)
        for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
            with Tapes(100 + len(out)) as t:
                en = pyq.EnergyAccumulator(mol, threshold=thr)(configs, wf)
            for k, v in en.items():
                out[f"{tag}_{thr_tag}_{k}"] = np.asarray(v)
            natm_ecp = sum(1 for a in mol._atom if a[0] in mol._ecp)
            N = sum(mol.nelec)
            out[f"{tag}_{thr_tag}_rot"] = np.asarray(t.log["rot"]).reshape(N, natm_ecp, 3, 3)
            out[f"{tag}_{thr_tag}_unif"] = np.asarray(t.log["random"]).reshape(N, natm_ecp, W)
        if tag == "h2o":  # one ecp_ea call in detail (eval_ecp.py:83-132)
            with Tapes(77) as t:
                d = eval_ecp.ecp_ea(mol, configs, wf, 1, mol._atom[0], 10.0)
            out["ea_rot"], out["ea_unif"] = t.log["rot"][0], t.log["random"][0]
            for k in ("total", "local", "mask", "ratio", "v_l", "P_l"):
                out["ea_" + k] = np.asarray(d[k])
            out["ea_epos"] = d["epos"].configs
        ee, ei, ii = refenergy.OpenCoulomb(mol).energy(configs)
        out[tag + "_ii"] = np.asarray(ii)
    save("g10_energy", **out)


# ------------------------------------------------------------------ G33 the ECP integrator's quadrature rules
def g_ecp_naip():
    """EnergyAccumulator(mol, naip=...) (accumulators.py:48-51 -> eval_ecp.ecp, eval_ecp.py:21-40, get_P_l :228-252) on a water
    molecule whose oxygen has s, p, d non-local channels: naip = None (12 points at O, 6 at H), 18, 26, 32, 50 and 6;
    deterministic mask (threshold -1) and the stochastic one (threshold 10)."""
    mol = systems.water_multichannel()
    mf = systems.random_mf(mol)
    wf = make_wf(mol, mf)
    W, N = 4, sum(mol.nelec)
    configs = walkers(mol, W, 33)
    rng = np.random.default_rng(5)
    configs.configs[:, :3, :] = mol.atom_coords()[0] + 0.4 * rng.standard_normal((W, 3, 3))  # electrons inside the oxygen core
    configs.configs[:, 5, :] = mol.atom_coords()[1] + 0.2 * rng.standard_normal((W, 3))     # and one at a hydrogen
    out = {"configs": configs.configs.copy()}
    wf.recompute(configs)
    natm_ecp = sum(1 for a in mol._atom if a[0] in mol._ecp)
    for naip in (None, 6, 18, 26, 32, 50):
        for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
            with Tapes(3300 + len(out)) as t:
                en = pyq.EnergyAccumulator(mol, threshold=thr, naip=naip)(configs, wf)
            tag = f"naip{naip}_{thr_tag}"
            out[tag + "_ecp"], out[tag + "_total"] = np.asarray(en["ecp"]), np.asarray(en["total"])
            out[tag + "_rot"] = np.asarray(t.log["rot"]).reshape(N, natm_ecp, 3, 3)
            out[tag + "_unif"] = np.asarray(t.log["random"]).reshape(N, natm_ecp, W)
    # s .. g non-local channels (the reference's Legendre table ends at l = 4, eval_ecp.py:203-225): six channels with the local one
    mol4 = systems.water_multichannel(lmax=4)
    wf4 = make_wf(mol4, systems.random_mf(mol4))
    wf4.recompute(configs)
    for naip in (None, 26, 50):
        with Tapes(3390 + len(out)) as t:
            en = pyq.EnergyAccumulator(mol4, threshold=10.0, naip=naip)(configs, wf4)
        tag = f"l4_naip{naip}"
        out[tag + "_ecp"], out[tag + "_total"] = np.asarray(en["ecp"]), np.asarray(en["total"])
        out[tag + "_rot"] = np.asarray(t.log["rot"]).reshape(N, natm_ecp, 3, 3)
        out[tag + "_unif"] = np.asarray(t.log["random"]).reshape(N, natm_ecp, W)
    for naip in (6, 12, 18, 26, 32, 50):  # the grids themselves (eval_ecp.py:278-336)
        pts, wts = eval_ecp.generate_quadrature_grids()[naip]
        out[f"grid{naip}_points"], out[f"grid{naip}_weights"] = pts, wts
    save("g33_ecp_naip", **out)


# ------------------------------------------------------------------ G34 the batched ECP integrator (jax_ecp.py)
def g_ecp_batched():
    """jax_ecp.ECPAccumulator (jax_ecp.py:22-142) — what EnergyAccumulator(use_old_ecp=False) evaluates — on the water molecule
    with s, p, d channels at the oxygen (12 + 6 + 6 points per electron): the default selection (12 deterministic + 1 random),
    6 + 3, and everything (24 + 0); its nonlocal_tmoves for two electrons; the EnergyAccumulator dict with use_old_ecp=False.
    numpy's argsort is made stable while the reference runs: the points of one atom share their probability, and which of
    them an unstable sort keeps where the cut falls inside an atom is an implementation detail of the sort (the fixture
    records how many walkers that concerns)."""
    import pyqmc.observables.jax_ecp as jax_ecp

    mol = systems.water_multichannel()
    mf = systems.random_mf(mol)
    wf = make_wf(mol, mf)
    W, N = 6, sum(mol.nelec)
    configs = walkers(mol, W, 34)
    rng = np.random.default_rng(6)
    configs.configs[:, :2, :] = mol.atom_coords()[0] + 0.4 * rng.standard_normal((W, 2, 3))
    configs.configs[:, 5, :] = mol.atom_coords()[1] + 0.15 * rng.standard_normal((W, 3))  # closer to a hydrogen than to the oxygen
    out = {"configs": configs.configs.copy()}
    wf.recompute(configs)
    real_argsort = np.argsort

    def stable_argsort(a, axis=-1, kind=None, order=None):
        return real_argsort(a, axis=axis, kind="stable", order=order)

    natom = mol.natm
    for tag, kws in (("default", {}), ("sel6_3", dict(nselect_deterministic=6, nselect_random=3)),
                     ("all", dict(nselect_deterministic=24, nselect_random=0)), ("fixedgrid", dict(stochastic_rotation=False))):
        acc = jax_ecp.ECPAccumulator(mol, **kws)
        nsr = acc.nselect_random
        out[tag + "_naip"], out[tag + "_nsd"], out[tag + "_nsr"] = np.asarray(acc.naip), acc.nselect_deterministic, nsr
        np.argsort = stable_argsort
        try:
            with Tapes(3400 + len(out)) as t:
                out[tag + "_ecp"] = np.asarray(acc(configs, wf))
            out[tag + "_rot"] = np.asarray(t.log["rot"]).reshape(N, natom, 3, 3)
            out[tag + "_unif"] = np.asarray(t.log["random"]).reshape(N, W, nsr) if t.log["random"] else np.zeros((N, W, 0))
            for e in (1, 5):
                with Tapes(3450 + e) as t:
                    d = acc.nonlocal_tmoves(configs, wf, e, 0.02)
                out[f"{tag}_tm{e}_rot"] = np.asarray(t.log["rot"]).reshape(natom, 3, 3)
                out[f"{tag}_tm{e}_unif"] = np.asarray(t.log["random"]).reshape(W, nsr) if t.log["random"] else np.zeros((W, 0))
                out[f"{tag}_tm{e}_ratio"], out[f"{tag}_tm{e}_weight"] = np.asarray(d["ratio"]), np.asarray(d["weight"])
                out[f"{tag}_tm{e}_epos"] = d["configs"].configs
        finally:
            np.argsort = real_argsort
        # the same draws under numpy's default sort: how far the reference's own answer depends on the tie order
        rots, unis = iter(out[tag + "_rot"].reshape(-1, 3, 3)), iter(out[tag + "_unif"])
        saved = (eval_ecp.scipy, np.random.random)
        ns = types.SimpleNamespace
        eval_ecp.scipy = ns(spatial=ns(transform=ns(Rotation=ns(random=lambda: ns(as_matrix=lambda R=None: next(rots))))))
        np.random.random = lambda size=None: next(unis)
        try:
            out[tag + "_ecp_default_sort"] = np.asarray(acc(configs, wf))
        finally:
            eval_ecp.scipy, np.random.random = saved
    np.argsort = stable_argsort
    try:
        with Tapes(3499) as t:
            en = pyq.EnergyAccumulator(mol, use_old_ecp=False)(configs, wf)
    finally:
        np.argsort = real_argsort
    for k, v in en.items():
        out["energy_" + k] = np.asarray(v)
    out["energy_rot"] = np.asarray(t.log["rot"]).reshape(N, natom, 3, 3)
    out["energy_unif"] = np.asarray(t.log["random"]).reshape(N, W, 1)
    save("g34_ecp_batched", **out)


# ------------------------------------------------------------------ G35 more than 64 electrons per spin
def g_big():
    """(H2O)18: 72 + 72 electrons, 414 AOs, 72 orbitals per spin (slater.py:155-260 takes any number of electrons) — the
    update / testvalue / recompute triangle for the first, the last and the two middle electrons (2 walkers), and one
    vmc_worker sweep with the energy (mc.py:102-153) of 2 walkers with every draw recorded."""
    mol = systems.water_cluster(3, 3, 2)
    mf = systems.random_mf(mol)
    wf = make_wf(mol, mf)
    out = {}
    protocol_dump("", mol, mf, wf, W=2, seed=35, electrons=[0, 71, 72, 143], out=out)
    for k in ("mo_coeff_alpha", "mo_coeff_beta", "acoeff"):  # (regenerated from the seeds by the tests; 0.5 MB each)
        out.pop(k)
    W, N, nsteps, tstep = 2, sum(mol.nelec), 1, 0.3
    natm_ecp = sum(1 for a in mol._atom if a[0] in mol._ecp)
    configs = walkers(mol, W, 135)
    out["vmc_start"] = configs.configs.copy()
    accepts = []
    orig_update = wf.updateinternals

    def spy(e, epos, cfg, mask=None, saved_values=None, _o=orig_update):
        accepts.append(np.asarray(mask).copy())
        return _o(e, epos, cfg, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(3500) as t:
        blk, configs = vmc_worker(wf, configs, tstep, nsteps, {"energy": pyq.EnergyAccumulator(mol)})
    wf.updateinternals = orig_update
    out["vmc_tstep"], out["vmc_nsteps"] = tstep, nsteps
    out["vmc_gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
    out["vmc_unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
    out["vmc_ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm_ecp, 3, 3)
    out["vmc_ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm_ecp, W)
    out["vmc_accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
    out["vmc_final"] = configs.configs.copy()
    out["vmc_final_log"] = wf.value()[1]
    for k, v in blk.items():
        if "time" not in k:
            out[f"vmc_blk_{k}"] = np.asarray(v)
    save("g35_big", **out)


# ------------------------------------------------------------------ G37 VMC trajectory of the headline system
def g_vmc_cluster():
    """(H2O)8, the 64-electron system of BASELINE.json's metric: one vmc_worker sweep (mc.py:102-153) of 4 walkers with the energy, every
    draw recorded — the reference-generated trajectory the resident sweep (k_sweep_r8: 32 electrons per spin, both orbital tiles, the
    K-split contraction) is held to directly (round-5 verdict, item 2)."""
    mol = systems.water_cluster()
    mf = systems.random_mf(mol)
    wf = make_wf(mol, mf)
    W, N, nsteps, tstep = 4, sum(mol.nelec), 1, 0.3
    natm_ecp = sum(1 for a in mol._atom if a[0] in mol._ecp)
    configs = walkers(mol, W, 137)
    out = {"start": configs.configs.copy()}
    accepts = []
    orig_update = wf.updateinternals

    def spy(e, epos, cfg, mask=None, saved_values=None, _o=orig_update):
        accepts.append(np.asarray(mask).copy())
        return _o(e, epos, cfg, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(3700) as t:
        blk, configs = vmc_worker(wf, configs, tstep, nsteps, {"energy": pyq.EnergyAccumulator(mol)})
    wf.updateinternals = orig_update
    out["tstep"], out["nsteps"] = tstep, nsteps
    out["gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
    out["unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
    out["ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm_ecp, 3, 3)
    out["ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm_ecp, W)
    out["accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
    out["final"] = configs.configs.copy()
    out["final_log"] = wf.value()[1]
    for k, v in blk.items():
        if "time" not in k:
            out[f"blk_{k}"] = np.asarray(v)
    for k in ("acoeff", "bcoeff"):
        out[k] = wf.wf_factors[1].parameters[k]
    print("cluster vmc: acceptance", blk["acceptance"], "E", blk["energytotal"])
    save("g37_vmc_cluster", **out)


# ------------------------------------------------------------------ G11 VMC trajectory
def g_vmc():
    out = {}
    for tag, mol, W, nsteps, tstep in (("h2o", systems.water(), 6, 3, 0.3), ("he", systems.helium(), 10, 4, 0.5)):
        mf = systems.random_mf(mol)
        wf = make_wf(mol, mf)
        N = sum(mol.nelec)
        natm_ecp = sum(1 for a in mol._atom if a[0] in mol._ecp)
        for attempt in range(20):
            configs = walkers(mol, W, 31 + attempt)
            start = configs.configs.copy()
            accepts = []
            orig_update = wf.updateinternals

            def spy(e, epos, cfg, mask=None, saved_values=None, _o=orig_update):
                accepts.append(np.asarray(mask).copy())
                return _o(e, epos, cfg, mask=mask, saved_values=saved_values)

            wf.updateinternals = spy
            with Tapes(500 + attempt) as t:
                blk, configs = vmc_worker(wf, configs, tstep, nsteps, {"energy": pyq.EnergyAccumulator(mol)})
            wf.updateinternals = orig_update
            break
        out[tag + "_start"] = start
        out[tag + "_tstep"], out[tag + "_nsteps"] = tstep, nsteps
        out[tag + "_gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
        out[tag + "_unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
        out[tag + "_ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm_ecp, 3, 3)
        out[tag + "_ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm_ecp, W)
        out[tag + "_accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
        out[tag + "_final"] = configs.configs.copy()
        s, l = wf.value()
        out[tag + "_final_log"] = l
        for k, v in blk.items():
            if "time" not in k:
                out[f"{tag}_blk_{k}"] = np.asarray(v)
        out[tag + "_blk_keys"] = np.asarray(sorted(blk.keys()))
        for k in ("acoeff", "bcoeff"):
            out[f"{tag}_{k}"] = wf.wf_factors[1].parameters[k]
    save("g11_vmc", **out)


# ------------------------------------------------------------------ G7 three-body Jastrow (+ multi-determinant: config C4 shape)
def g_jastrow3():
    import pyqmc.wftools as wftools

    out = {}
    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=6)
    dets = systems.random_determinants(mol, mf, 12)
    out["det_json"] = np.asarray(repr(dets))
    wf, _ = pyq.generate_wf(mol, mf, jastrow=[pyq.generate_jastrow, wftools.generate_jastrow3], jastrow_kws=[{}, {}],
                            slater_kws=dict(evaluate_orbitals_with="numba", determinants=dets))
    rng = np.random.default_rng(11)
    j2, j3 = wf.wf_factors[1], wf.wf_factors[2]
    j2.parameters["acoeff"] = 0.05 * rng.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * rng.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    j3.parameters["ccoeff"] = 0.1 * np.random.default_rng(12).standard_normal(j3.parameters["ccoeff"].shape)
    out["ccoeff"] = j3.parameters["ccoeff"].copy()
    W = 5
    configs = walkers(mol, W, 71)
    out["configs"] = configs.configs.copy()
    names = {"j3": j3, "wf": wf}
    for nm, w in names.items():
        s_, l_ = w.recompute(configs)
        out[f"{nm}_recompute_log"] = l_
    prng = np.random.default_rng(13)
    out["electrons"] = np.asarray([0, 3, 4, 7])
    for e in out["electrons"]:
        e = int(e)
        newpos = configs.configs[:, e, :] + 0.3 * prng.standard_normal((W, 3))
        aux = configs.configs[:, e, None, :] + 0.4 * prng.standard_normal((W, 6, 3))
        mask = prng.random(W) > 0.35
        mask[0] = True
        accept = prng.random(W) > 0.4
        out[f"e{e}_newpos"], out[f"e{e}_aux"], out[f"e{e}_mask"], out[f"e{e}_accept"] = newpos, aux, mask, accept
        ep, ea = configs.make_irreducible(e, newpos), configs.make_irreducible(e, aux)
        for nm, w in names.items():
            p = f"e{e}_{nm}_"
            g, v, _ = w.gradient_value(e, ep)
            out[p + "gv_grad"], out[p + "gv_val"] = g, v
            out[p + "grad"] = w.gradient(e, ep)
            g, l = w.gradient_laplacian(e, ep)
            out[p + "gl_grad"], out[p + "gl_lap"] = g, l
            out[p + "testvalue"] = w.testvalue(e, ep)[0]
            out[p + "testvalue_aux"] = w.testvalue(e, ea, mask)[0]
        # NOTE reference defect: ThreeBodyJastrow.updateinternals (three_body_jastrow.py:149-189) takes the OLD
        # position of electron e from its `configs` argument, but vmc_worker/dmc_propagate move `configs` first
        # (mc.py:135-136), which leaves its P_i sums stale (6.6e-3 error in log psi here).  The factor is
        # self-consistent (4e-16 vs recompute) when updated BEFORE the move, so the goldens use that order.
        wf.updateinternals(e, ep, configs, mask=accept)
        configs.move(e, ep, accept)
        for nm, w in names.items():
            out[f"e{e}_{nm}_post_log"] = w.value()[1]
    out["final_configs"] = configs.configs.copy()
    for nm, w in names.items():
        out[f"{nm}_final_recompute_log"] = w.recompute(configs)[1]
    # energy + short VMC trajectory of the full C4-shaped wave function
    with Tapes(300) as t:
        en = pyq.EnergyAccumulator(mol)(configs, wf)
    for k, v in en.items():
        out["energy_" + k] = np.asarray(v)
    out["energy_rot"] = np.asarray(t.log["rot"]).reshape(8, 3, 3, 3)
    out["energy_unif"] = np.asarray(t.log["random"]).reshape(8, 3, W)
    start = walkers(mol, W, 72)
    out["vmc_start"] = start.configs.copy()
    accepts = []
    orig = wf.updateinternals
    shadow = {"x": start.configs.copy()}  # walkers as they were before the driver's configs.move

    def spy(e, ep, c, mask=None, saved_values=None):
        accepts.append(np.asarray(mask).copy())
        moved = c.configs[:, e].copy()
        c.configs[:, e] = shadow["x"][:, e]  # present the un-moved walkers to updateinternals (see NOTE above)
        orig(e, ep, c, mask=mask, saved_values=saved_values)
        c.configs[:, e] = moved
        shadow["x"][:, e] = moved

    wf.updateinternals = spy
    with Tapes(301) as t:
        blk, cfg = vmc_worker(wf, start, 0.3, 2, {"energy": pyq.EnergyAccumulator(mol)})
    wf.updateinternals = orig
    out["vmc_gauss"] = np.asarray(t.log["normal"]).reshape(2, 8, W, 3)
    out["vmc_unif"] = np.asarray(t.log["rand"]).reshape(2, 8, W)
    out["vmc_ecp_rot"] = np.asarray(t.log["rot"]).reshape(2, 8, 3, 3, 3)
    out["vmc_ecp_unif"] = np.asarray(t.log["random"]).reshape(2, 8, 3, W)
    out["vmc_accepts"] = np.asarray(accepts).reshape(2, 8, W)
    out["vmc_final"] = cfg.configs.copy()
    for k, v in blk.items():
        if "time" not in k:
            out["vmc_blk_" + k] = np.asarray(v)
    save("g7_jastrow3_multidet", **out)



# ------------------------------------------------------------------ G13 periodic containers, minimal image, PBC Jastrow
def pbc_protocol(prefix, cell, names, W, seed, electrons, out, update_first=False, naip=6):
    """protocol_dump for wave-function factors living on PeriodicConfigs.  ``update_first``: see the NOTE in
    g_jastrow3 (needed whenever a ThreeBodyJastrow is among the factors)."""
    from pyqmc.configurations.coord import PeriodicConfigs

    rng = np.random.default_rng(seed)
    start = systems.initial_guess(cell, W, rng=np.random.default_rng(seed)).configs
    configs = PeriodicConfigs(start.copy(), cell.lattice_vectors())
    out[prefix + "configs"], out[prefix + "wrap"] = configs.configs.copy(), configs.wrap.copy()
    for nm, w in names.items():
        out[f"{prefix}{nm}_recompute_sign"], out[f"{prefix}{nm}_recompute_log"] = w.recompute(configs)
    out[prefix + "electrons"] = np.asarray(electrons)
    top = list(names.values())[-1]
    for e in electrons:
        newpos = configs.configs[:, e, :] + 0.6 * rng.standard_normal((W, 3))
        aux = configs.configs[:, e, None, :] + 0.8 * rng.standard_normal((W, naip, 3))
        mask = rng.random(W) > 0.35
        mask[0] = True
        accept = rng.random(W) > 0.4
        accept[-1] = True  # (the reference's pure-Python AO path cannot take an empty point list)
        out[f"{prefix}e{e}_newpos"], out[f"{prefix}e{e}_aux"] = newpos, aux
        out[f"{prefix}e{e}_mask"], out[f"{prefix}e{e}_accept"] = mask, accept
        ep = configs.make_irreducible(e, newpos)
        ea = configs.make_irreducible(e, aux, mask)
        out[f"{prefix}e{e}_ep_configs"], out[f"{prefix}e{e}_ep_wrap"] = ep.configs.copy(), ep.wrap.copy()
        out[f"{prefix}e{e}_ea_configs"], out[f"{prefix}e{e}_ea_wrap"] = ea.configs.copy(), ea.wrap.copy()
        for nm, w in names.items():
            p = f"{prefix}e{e}_{nm}_"
            g, v, _ = w.gradient_value(e, ep)
            out[p + "gv_grad"], out[p + "gv_val"] = g, v
            out[p + "grad"] = w.gradient(e, ep)
            g, l = w.gradient_laplacian(e, ep)
            out[p + "gl_grad"], out[p + "gl_lap"] = g, l
            g, l = w.gradient_laplacian(e, configs.electron(e))
            out[p + "gl0_grad"], out[p + "gl0_lap"] = g, l
            out[p + "testvalue"] = w.testvalue(e, ep)[0]
            out[p + "testvalue_mask"] = w.testvalue(e, ep, mask)[0]
            out[p + "testvalue_aux"] = w.testvalue(e, ea, mask)[0]
        if update_first:
            top.updateinternals(e, ep, configs, mask=accept)
            configs.move(e, ep, accept)
        else:
            configs.move(e, ep, accept)
            top.updateinternals(e, ep, configs, mask=accept)
        for nm, w in names.items():
            out[f"{prefix}e{e}_{nm}_post_sign"], out[f"{prefix}e{e}_{nm}_post_log"] = w.value()
    out[prefix + "final_configs"], out[prefix + "final_wrap"] = configs.configs.copy(), configs.wrap.copy()
    for nm, w in names.items():
        out[f"{prefix}{nm}_final_recompute_sign"], out[f"{prefix}{nm}_final_recompute_log"] = w.recompute(configs)


def g_pbc():
    import pyqmc.wftools as wftools
    from pyqmc.configurations.coord import PeriodicConfigs
    from pyqmc.configurations.distance import MinimalImageDistance
    from pyqmc.pbc.pbc import enforce_pbc

    out = {}
    # the known-answer table of the reference's tests/unit/test_pbcs.py:19-72 (inputs only; outputs from the call)
    tri = np.array([[1.2, 0, 0], [0.6, 1.2 * np.sqrt(3) / 2, 0], [0, 0, 0.8]])
    trans = np.array([[0.1, 0.1, 0.1], [1.3, 0, 0.2], [0.9, 1.8 * np.sqrt(3) / 2, 0], [0, 0, 1.1],
                      [2.34, 1.35099963, 0], [0.48, 1.24707658, 0], [-2.52, 2.28630707, -0.32]]) + 1e-14
    out["tri_lat"], out["tri_in"] = tri, trans
    out["tri_pos"], out["tri_wrap"] = enforce_pbc(tri, trans)
    rng = np.random.default_rng(131)
    rot = scipy.spatial.transform.Rotation.from_rotvec([0.3, -0.5, 0.4]).as_matrix()
    lats = {"diag": np.diag([2.0, 3.0, 4.0]), "ortho": np.diag([2.0, 3.0, 4.0]) @ rot,
            "general": systems.diamond_primitive().lattice_vectors(), "tri": tri}
    for tag, lat in lats.items():
        x = rng.standard_normal((5, 6, 3)) * 3.0
        pos, wrap = enforce_pbc(lat, x)
        mid = MinimalImageDistance(lat)
        vec = pos[:, 0] + 0.7 * rng.standard_normal((5, 3))
        out[f"mi_{tag}_lat"], out[f"mi_{tag}_x"], out[f"mi_{tag}_pos"], out[f"mi_{tag}_wrap"] = lat, x, pos, wrap
        out[f"mi_{tag}_vec"] = vec
        out[f"mi_{tag}_dist_i"] = mid.dist_i(pos, vec)
        out[f"mi_{tag}_dist_matrix"] = mid.dist_matrix(pos)[0]
        out[f"mi_{tag}_pairwise"] = mid.pairwise(pos[:, :2], pos[:, 2:])
        cfg = PeriodicConfigs(x.copy(), lat)
        aux = x[:, 1, None, :] + rng.standard_normal((5, 4, 3))
        m = rng.random((5, 4)) > 0.3
        el = cfg.make_irreducible(1, aux, m)
        out[f"mi_{tag}_aux"], out[f"mi_{tag}_auxmask"] = aux, m
        out[f"mi_{tag}_aux_pos"], out[f"mi_{tag}_aux_wrap"] = el.configs, el.wrap
    save("g13_pbc", **out)

    # Jastrow factors under PBC: cubic diamond (diagonal lattice, default rcut = min pi/|b_i|, wftools.py:82-83) and
    # the primitive fcc cell (27-image rule) with a cut-off beyond half the plane spacing
    out = {}
    for tag, cell, W, electrons, kws in (("cubic", systems.diamond_cubic(), 4, [0, 9, 16, 31], {}),
                                         ("prim", systems.diamond_primitive(), 6, [0, 3, 4, 7], {"rcut": 4.0})):
        rng = np.random.default_rng(17)
        j2, _ = wftools.generate_jastrow(cell, **kws)
        j2.parameters["acoeff"] = 0.05 * rng.standard_normal(j2.parameters["acoeff"].shape)
        b = 0.05 * rng.standard_normal(j2.parameters["bcoeff"].shape)
        b[0] = [-0.25, -0.5, -0.25]
        j2.parameters["bcoeff"] = b
        out[f"{tag}_acoeff"], out[f"{tag}_bcoeff"] = j2.parameters["acoeff"], j2.parameters["bcoeff"]
        out[f"{tag}_rcut"] = np.asarray(j2.a_basis.rcut)
        pbc_protocol(f"{tag}_", cell, {"jastrow": j2}, W, 41, electrons, out)
    cell = systems.diamond_primitive()
    j3, _ = wftools.generate_jastrow3(cell, rcut=3.0)
    j3.parameters["ccoeff"] = 0.1 * np.random.default_rng(12).standard_normal(j3.parameters["ccoeff"].shape)
    out["prim3_ccoeff"] = j3.parameters["ccoeff"].copy()
    pbc_protocol("prim3_", cell, {"j3": j3}, 5, 43, [1, 6], out, update_first=True)
    save("g14_pbc_jastrow", **out)



# ------------------------------------------------------------------ G15 periodic orbitals + Slater (zero twist, real phases)
def ref_pbc_objects(supercell, kpts, mo_coeff, Ls, determinants=None, precision=1e-2):
    """The reference's PeriodicAtomicOrbitalEvaluator / PBCOrbitalEvaluatorKpoints / Slater for a supercell, built by
    setting the attributes their constructors would set (pbcgto.py:596-636, orbitals.py:141-184, slater.py:181-225)
    because those constructors call pyscf (lattice-sum list, SCF conversion).  ``Ls`` replaces
    ``cell.get_lattice_Ls`` (sorted by norm as pbcgto.py:603 does); everything downstream is the reference's code."""
    import pyqmc.wf.numba.pbcgto as pbcgto
    import pyqmc.wf.orbitals as orbitals
    import pyqmc.wf.slater as rslater
    from pyqmc.configurations.distance import RawDistance
    from pyqmc.wf import determinant_tools

    prim = supercell.original_cell
    ev = object.__new__(pbcgto.PeriodicAtomicOrbitalEvaluator)
    refgto.AtomicOrbitalEvaluator.__init__(ev, prim)
    ev.kpts, ev.Ls = kpts, Ls
    ev.num_Ls, ev.atom_cutoff, ev.l_cutoff = pbcgto.max_Ls(ev.Ls, prim.lattice_vectors(), ev.basis_ls, ev.basis_arrays,
                                                            ev.splits, ev.l_splits, expcutoff=-3.5 * np.log(precision))
    ev.Lmax = ev.num_Ls.max()
    ev.phases = np.real_if_close(np.exp(1j * ev.Ls @ kpts.T))
    ev.dtype = ev.phases.dtype
    cplx = ev.dtype == complex or any(np.iscomplexobj(m) for sp in mo_coeff for m in sp)
    ev.get_wrapphase = orbitals.get_wrapphase_complex if ev.dtype == complex else orbitals.get_wrapphase_real
    ev.dist = RawDistance()
    ev._gto_func = dict(GTOval_sph=pbcgto._pbc_eval_gto, GTOval_sph_deriv1=pbcgto._pbc_eval_gto_grad,
                        GTOval_sph_deriv2=pbcgto._pbc_eval_gto_lap)
    oe = object.__new__(orbitals.PBCOrbitalEvaluatorKpoints)
    oe._cell, oe.S, oe.Lprim = prim, np.asarray(supercell.S), prim.lattice_vectors()
    oe._kpts = kpts
    oe.isgamma = np.abs(kpts).sum() < 1e-9
    nelec_per_kpt = [np.asarray([m.shape[1] for m in mo]) for mo in mo_coeff]
    oe.param_split = [np.cumsum(nelec_per_kpt[spin]) for spin in [0, 1]]
    oe.parm_names = ["mo_coeff_alpha", "mo_coeff_beta"]
    oe.parameters = {"mo_coeff_alpha": np.concatenate(mo_coeff[0], axis=1), "mo_coeff_beta": np.concatenate(mo_coeff[1], axis=1)}
    oe.mo_dtype = complex if cplx else float  # orbitals.py:161-166
    oe.get_wrapphase = orbitals.get_wrapphase_complex if cplx else orbitals.get_wrapphase_real
    oe.eval_gto = ev.eval_gto
    if determinants is None:
        determinants = [(1.0, [list(range(supercell.nelec[0])), list(range(supercell.nelec[1]))])]
    sl = object.__new__(rslater.Slater)
    sl.tol, sl._mol, sl._nelec = -1, supercell, supercell.nelec
    sl.myparameters = {}
    sl.myparameters["det_coeff"], sl._det_occup, sl._det_map = determinant_tools.create_packed_objects(determinants, tol=-1)
    sl.orbitals = oe
    sl.parameters = rslater.JoinParameters([sl.myparameters, oe.parameters])
    sl.dtype, sl.get_phase = (complex, rslater.get_complex_phase) if cplx else (float, np.sign)  # slater.py:212-216
    sl._gtoval, sl._gtoval_deriv1, sl._gtoval_deriv2 = "GTOval_sph", "GTOval_sph_deriv1", "GTOval_sph_deriv2"
    return ev, oe, sl


PBC_SLATER_CASES = {
    "gamma": (np.eye(3), 4, [0, 3, 4, 7]),
    "fcc2cubic": (np.array([[-1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 1.0, -1.0]]), 3, [0, 15, 16, 31]),
    "k222": (2.0 * np.eye(3), 2, [0, 31, 32, 63]),
}


def g_pbc_slater():
    import pyqmc.wftools as wftools
    from pyqmc.configurations.coord import PeriodicConfigs
    from pyqmc_amd import pbc as mypbc

    prim = systems.diamond_primitive()
    out = {}
    for tag, (S, W, electrons) in PBC_SLATER_CASES.items():
        sup = mypbc.get_supercell(prim, S)
        mf = mypbc.random_kmf(sup)
        # lattice translations of the PRIMITIVE cell, generously beyond the largest cut-off (stand-in for get_lattice_Ls)
        Ls = mypbc.lattice_points_within(prim.lattice_vectors(), 30.0)
        ev, oe, sl = ref_pbc_objects(sup, mf.kpts, mf.mo_coeff, Ls)
        out[f"{tag}_kpts"], out[f"{tag}_Ls"], out[f"{tag}_num_Ls"] = mf.kpts, Ls, ev.num_Ls
        out[f"{tag}_atom_cut"], out[f"{tag}_shell_cut"] = ev.atom_cutoff, ev.l_cutoff
        out[f"{tag}_atoms"] = sup.atom_coords()
        rng = np.random.default_rng(91)
        if tag != "k222":  # AO / MO level, at points both inside and outside the supercell
            pts = PeriodicConfigs((rng.random((1, 10, 3)) * 3 - 1) @ sup.lattice_vectors(), sup.lattice_vectors())
            out[f"{tag}_pts"], out[f"{tag}_pts_wrap"] = pts.configs.copy(), pts.wrap.copy()
            for nm, es in (("val", "GTOval_sph"), ("grad", "GTOval_sph_deriv1"), ("lap", "GTOval_sph_deriv2")):
                ao = oe.aos(es, pts)
                out[f"{tag}_ao_{nm}"] = ao
                out[f"{tag}_mo_{nm}"] = oe.mos(ao, 0)
            # is the reference's num_Ls truncation a superset of what the cut-offs let through?
            full = np.full_like(ev.num_Ls, len(Ls))
            keep, ev.num_Ls = ev.num_Ls, full
            ev.Lmax = len(Ls)
            out[f"{tag}_ao_lap_allLs"] = oe.aos("GTOval_sph_deriv2", pts)
            ev.num_Ls, ev.Lmax = keep, keep.max()
            print(tag, "num_Ls", keep, "effect of the num_Ls truncation:",
                  np.abs(out[f"{tag}_ao_lap_allLs"] - out[f"{tag}_ao_lap"]).max())
        j2, _ = wftools.generate_jastrow(sup)
        jr = np.random.default_rng(17)
        j2.parameters["acoeff"] = 0.05 * jr.standard_normal(j2.parameters["acoeff"].shape)
        b = 0.05 * jr.standard_normal(j2.parameters["bcoeff"].shape)
        b[0] = [-0.25, -0.5, -0.25]
        j2.parameters["bcoeff"] = b
        from pyqmc.wf.multiplywf import MultiplyWF

        wf = MultiplyWF(sl, j2)
        pbc_protocol(f"{tag}_", sup, {"slater": sl, "jastrow": j2, "wf": wf}, W, 51, electrons, out)
    save("g15_pbc_orbitals", **out)



# ------------------------------------------------------------------ G16 periodic energies (Ewald, ECP, kinetic) + VMC trajectory
def ref_pbc_wf(tag):
    import pyqmc.wftools as wftools
    from pyqmc.wf.multiplywf import MultiplyWF
    from pyqmc_amd import pbc as mypbc

    prim = systems.diamond_primitive()
    sup = mypbc.get_supercell(prim, PBC_SLATER_CASES[tag][0])
    mf = mypbc.random_kmf(sup)
    Ls = mypbc.lattice_points_within(prim.lattice_vectors(), 30.0)
    _, _, sl = ref_pbc_objects(sup, mf.kpts, mf.mo_coeff, Ls)
    j2, _ = wftools.generate_jastrow(sup)
    jr = np.random.default_rng(17)
    j2.parameters["acoeff"] = 0.05 * jr.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * jr.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    return sup, MultiplyWF(sl, j2)


def g_pbc_energy():
    import pyqmc.observables.ewald as refewald
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    for tag, W in (("gamma", 4), ("fcc2cubic", 2)):
        sup, wf = ref_pbc_wf(tag)
        N, natm_ecp = sum(sup.nelec), sup.natm
        configs = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(61)).configs.copy(), sup.lattice_vectors())
        out[f"{tag}_configs"] = configs.configs.copy()
        ew = refewald.Ewald(sup, ewald_gmax=200 if tag == "gamma" else 10)
        ee, ei, ii = ew.energy(configs)
        out[f"{tag}_ewald_ee"], out[f"{tag}_ewald_ei"], out[f"{tag}_ewald_ii"] = ee, ei, np.asarray(ii)
        out[f"{tag}_ewald_alpha"], out[f"{tag}_ewald_ng"] = np.asarray(ew.alpha), np.asarray(len(ew.gweight))
        out[f"{tag}_ewald_gpoints"], out[f"{tag}_ewald_gweight"] = np.asarray(ew.gpoints), np.asarray(ew.gweight)
        wf.recompute(configs)
        for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
            with Tapes(700 + len(out)) as t:
                en = pyq.EnergyAccumulator(sup, threshold=thr, ewald_gmax=10)(configs, wf)
            for k, v in en.items():
                out[f"{tag}_{thr_tag}_{k}"] = np.asarray(v)
            out[f"{tag}_{thr_tag}_rot"] = np.asarray(t.log["rot"]).reshape(N, natm_ecp, 3, 3)
            out[f"{tag}_{thr_tag}_unif"] = np.asarray(t.log["random"]).reshape(N, natm_ecp, W)
        if tag == "gamma":  # VMC trajectory with energies, walkers crossing the cell boundary
            nsteps, tstep = 2, 0.5
            start = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(62)).configs.copy(), sup.lattice_vectors())
            out["vmc_start"], out["vmc_start_wrap"] = start.configs.copy(), start.wrap.copy()
            accepts = []
            orig = wf.updateinternals

            def spy(e, epos, cfg, mask=None, saved_values=None):
                accepts.append(np.asarray(mask).copy())
                return orig(e, epos, cfg, mask=mask, saved_values=saved_values)

            wf.updateinternals = spy
            with Tapes(801) as t:
                blk, cfg = vmc_worker(wf, start, tstep, nsteps, {"energy": pyq.EnergyAccumulator(sup, ewald_gmax=10)})
            wf.updateinternals = orig
            out["vmc_tstep"], out["vmc_nsteps"] = tstep, nsteps
            out["vmc_gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
            out["vmc_unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
            out["vmc_ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm_ecp, 3, 3)
            out["vmc_ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm_ecp, W)
            out["vmc_accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
            out["vmc_final"], out["vmc_final_wrap"] = cfg.configs.copy(), cfg.wrap.copy()
            out["vmc_final_log"] = wf.value()[1]
            for k, v in blk.items():
                if "time" not in k:
                    out["vmc_blk_" + k] = np.asarray(v)
            print("wrap moved:", np.abs(cfg.wrap - out["vmc_start_wrap"]).sum(), "acceptance", blk["acceptance"])
    save("g16_pbc_energy", **out)



# ------------------------------------------------------------------ G17 periodic DMC propagate (T-moves, Ewald, wrap counters)
def g_pbc_dmc():
    import pyqmc.method.dmc as refdmc
    import pyqmc.wf.orbitals as reforb
    from pyqmc.configurations.coord import PeriodicConfigs

    # like g_dmc: the un-JIT-ed AO path cannot take an empty point list (an all-False T-move mask)
    orig_aos = reforb.PBCOrbitalEvaluatorKpoints.aos

    def aos(self, eval_str, configs, mask=None):
        coords = configs.configs if mask is None else configs.configs[mask]
        if coords.size == 0:
            nao, nk = self.parameters["mo_coeff_alpha"].shape[0], len(self._kpts)
            comp = () if "deriv" not in eval_str else ((4,) if "deriv1" in eval_str else (5,))
            return np.zeros((nk, *comp, *coords.shape[:-1], nao))
        return orig_aos(self, eval_str, configs, mask)

    reforb.PBCOrbitalEvaluatorKpoints.aos = aos
    out = {}
    sup, wf = ref_pbc_wf("gamma")
    W, nsteps, tstep = 5, 2, 0.1
    rng = np.random.default_rng(4)
    configs = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(63)).configs.copy(), sup.lattice_vectors())
    # pull electrons close to the carbon cores so the ECP mask passes and T-moves get accepted
    near = sup.atom_coords()[[0, 1, 0, 1]][None] + 0.25 * rng.standard_normal((W, 4, 3))
    configs = PeriodicConfigs(np.concatenate([near, configs.configs[:, 4:]], axis=1)[:, [0, 4, 1, 5, 2, 6, 3, 7]], sup.lattice_vectors())
    out["start"], out["start_wrap"] = configs.configs.copy(), configs.wrap.copy()
    weights = 1.0 + 0.1 * rng.standard_normal(W)
    out["weights0"] = weights.copy()
    e_trial, e_est, branchcut = -10.0, -10.2, 3.0
    out["params"] = np.array([tstep, branchcut, e_trial, e_est, nsteps])
    accepts = []
    orig = wf.updateinternals

    def spy(e, epos, cfg, mask=None, saved_values=None):
        accepts.append(np.asarray(mask).copy())
        return orig(e, epos, cfg, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(950) as t:
        df, configs, weights = refdmc.dmc_propagate(wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps=nsteps,
                                                    accumulators={"energy": pyq.EnergyAccumulator(sup, ewald_gmax=10)})
    wf.updateinternals = orig
    reforb.PBCOrbitalEvaluatorKpoints.aos = orig_aos
    out["normal"] = np.asarray(t.log["normal"])
    rand = t.log["rand"]
    out["rand_scalar"] = np.asarray([float(r) for r in rand if np.ndim(r) == 0])
    out["rand_vector"] = np.asarray([r for r in rand if np.ndim(r) == 1])
    out["rot"] = np.asarray(t.log["rot"])
    out["random"] = np.asarray(t.log["random"])
    out["accepts"] = np.asarray(accepts)
    out["final"], out["final_wrap"] = configs.configs.copy(), configs.wrap.copy()
    out["weights"] = weights
    for k, v in df.items():
        out["df_" + k] = np.asarray(v)
    out["df_keys"] = np.asarray(sorted(df.keys()))
    print("pbc dmc: wrap moved", np.abs(out["final_wrap"] - out["start_wrap"]).sum(), "tmove acc", df.get("tmove_acceptance"))
    save("g17_pbc_dmc", **out)



# ------------------------------------------------------------------ G18 testvalue_many (density-matrix accumulators)
def g_testvalue_many():
    import pyqmc.wftools as wftools
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    # open: 12-determinant H2O, Slater x two-body x three-body
    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=6)
    dets = systems.random_determinants(mol, mf, 12)
    wf, _ = pyq.generate_wf(mol, mf, jastrow=[pyq.generate_jastrow, wftools.generate_jastrow3], jastrow_kws=[{}, {}],
                            slater_kws=dict(evaluate_orbitals_with="numba", determinants=dets))
    rng = np.random.default_rng(11)
    j2, j3 = wf.wf_factors[1], wf.wf_factors[2]
    j2.parameters["acoeff"] = 0.05 * rng.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * rng.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    j3.parameters["ccoeff"] = 0.1 * np.random.default_rng(12).standard_normal(j3.parameters["ccoeff"].shape)
    W = 6
    configs = walkers(mol, W, 81)
    out["h2o_configs"] = configs.configs.copy()
    wf.recompute(configs)
    prng = np.random.default_rng(82)
    aux = prng.standard_normal((W, 3)) * 1.5
    mask = prng.random(W) > 0.3
    mask[0] = True
    es = np.array([0, 3, 4, 7, 5])
    out["h2o_aux"], out["h2o_mask"], out["h2o_es"] = aux, mask, es
    epos = configs.make_irreducible(0, aux)
    # parameter gradients of the same state (slater.py:462-542, jastrowspin.py:457-464, three_body_jastrow.py:657-719)
    pg = wf.pgradient()
    for k in pg.keys():
        out["h2o_pgrad_" + k] = np.asarray(pg[k])
    out["h2o_pgrad_keys"] = np.asarray(sorted(pg.keys()))
    for nm, w in (("slater", wf.wf_factors[0]), ("j2", j2), ("j3", j3), ("wf", wf)):
        out[f"h2o_{nm}"] = w.testvalue_many(es, epos)
        # (no masked goldens: the reference's testvalue_many allocate for all walkers and fail on a real mask,
        #  slater.py:454, jastrowspin.py:438; its accumulators never pass one)
    # periodic: diamond 4-fold supercell, Slater x two-body
    sup, pwf = ref_pbc_wf("fcc2cubic")
    W = 3
    cfg = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(83)).configs.copy(), sup.lattice_vectors())
    out["pbc_configs"], out["pbc_wrap"] = cfg.configs.copy(), cfg.wrap.copy()
    pwf.recompute(cfg)
    aux = (prng.random((W, 3)) * 3 - 1) @ sup.lattice_vectors()
    es = np.array([1, 15, 16, 30])
    out["pbc_aux"], out["pbc_es"] = aux, es
    epos = cfg.make_irreducible(0, aux)
    for nm, w in (("slater", pwf.wf_factors[0]), ("j2", pwf.wf_factors[1]), ("wf", pwf)):
        out[f"pbc_{nm}"] = w.testvalue_many(es, epos)
    save("g18_testvalue_many", **out)



# ------------------------------------------------------------------ G19 complex Bloch orbitals (k-points off the TRIM set, zero twist)
def ref_pbc_wf_complex():
    import pyqmc.wftools as wftools
    from pyqmc.wf.multiplywf import MultiplyWF
    from pyqmc_amd import pbc as mypbc

    prim = systems.diamond_primitive()
    sup = mypbc.get_supercell(prim, np.diag([3.0, 1.0, 1.0]))
    mf = mypbc.random_kmf(sup, complex_coeff=True)
    Ls = mypbc.lattice_points_within(prim.lattice_vectors(), 30.0)
    ev, oe, sl = ref_pbc_objects(sup, mf.kpts, mf.mo_coeff, Ls)
    j2, _ = wftools.generate_jastrow(sup)
    jr = np.random.default_rng(17)
    j2.parameters["acoeff"] = 0.05 * jr.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * jr.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    return sup, mf, Ls, oe, sl, j2, MultiplyWF(sl, j2)


def g_pbc_complex():
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    sup, mf, Ls, oe, sl, j2, wf = ref_pbc_wf_complex()
    assert sl.dtype == complex
    out["kpts"], out["Ls"], out["atoms"] = mf.kpts, Ls, sup.atom_coords()
    rng = np.random.default_rng(93)
    pts = PeriodicConfigs((rng.random((1, 8, 3)) * 3 - 1) @ sup.lattice_vectors(), sup.lattice_vectors())
    out["pts"] = pts.configs.copy()
    for nm, es in (("val", "GTOval_sph"), ("lap", "GTOval_sph_deriv2")):
        ao = oe.aos(es, pts)
        out[f"ao_{nm}"] = ao
        out[f"mo_{nm}"] = oe.mos(ao, 0)
    pbc_protocol("", sup, {"slater": sl, "jastrow": j2, "wf": wf}, 3, 53, [0, 5, 12, 23], out)
    # energies (complex ecp / total) and a VMC trajectory
    W, N, natm = 3, sum(sup.nelec), sup.natm
    cfg = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(64)).configs.copy(), sup.lattice_vectors())
    out["en_configs"] = cfg.configs.copy()
    wf.recompute(cfg)
    for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
        with Tapes(720 + len(thr_tag)) as t:
            en = pyq.EnergyAccumulator(sup, threshold=thr, ewald_gmax=10)(cfg, wf)
        for k, v in en.items():
            out[f"en_{thr_tag}_{k}"] = np.asarray(v)
        out[f"en_{thr_tag}_rot"] = np.asarray(t.log["rot"]).reshape(N, natm, 3, 3)
        out[f"en_{thr_tag}_unif"] = np.asarray(t.log["random"]).reshape(N, natm, W)
    nsteps, tstep = 2, 0.5
    start = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(65)).configs.copy(), sup.lattice_vectors())
    out["vmc_start"], out["vmc_start_wrap"] = start.configs.copy(), start.wrap.copy()
    accepts = []
    orig = wf.updateinternals

    def spy(e, epos, c, mask=None, saved_values=None):
        accepts.append(np.asarray(mask).copy())
        return orig(e, epos, c, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(802) as t:
        blk, cfg2 = vmc_worker(wf, start, tstep, nsteps, {"energy": pyq.EnergyAccumulator(sup, ewald_gmax=10)})
    wf.updateinternals = orig
    out["vmc_tstep"], out["vmc_nsteps"] = tstep, nsteps
    out["vmc_gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
    out["vmc_unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
    out["vmc_ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm, 3, 3)
    out["vmc_ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm, W)
    out["vmc_accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
    out["vmc_final"], out["vmc_final_wrap"] = cfg2.configs.copy(), cfg2.wrap.copy()
    out["vmc_final_sign"], out["vmc_final_log"] = wf.value()
    for k, v in blk.items():
        if "time" not in k:
            out["vmc_blk_" + k] = np.asarray(v)
    save("g19_pbc_complex", **out)



# ------------------------------------------------------------------ G20 twisted boundary conditions (complex AOs, wrap phases)
TWIST_CASES = {"prim": (np.eye(3), (0.25, 0.1, -0.3), 4, [0, 3, 4, 7]), "s211": (np.diag([2.0, 1.0, 1.0]), (0.2, -0.15, 0.4), 3, [0, 7, 8, 15])}


def ref_twisted_wf(tag):
    import pyqmc.wftools as wftools
    from pyqmc.wf.multiplywf import MultiplyWF
    from pyqmc_amd import pbc as mypbc

    S, twist, W, electrons = TWIST_CASES[tag]
    prim = systems.diamond_primitive()
    sup = mypbc.get_supercell(prim, S)
    mf = mypbc.random_kmf(sup, complex_coeff=True, twist=twist)
    Ls = mypbc.lattice_points_within(prim.lattice_vectors(), 30.0)
    ev, oe, sl = ref_pbc_objects(sup, mf.kpts, mf.mo_coeff, Ls)
    j2, _ = wftools.generate_jastrow(sup)
    jr = np.random.default_rng(17)
    j2.parameters["acoeff"] = 0.05 * jr.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * jr.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    return sup, mf, Ls, oe, sl, j2, MultiplyWF(sl, j2), W, electrons


def g_pbc_twist():
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    for tag in TWIST_CASES:
        sup, mf, Ls, oe, sl, j2, wf, W, electrons = ref_twisted_wf(tag)
        out[f"{tag}_kpts"], out[f"{tag}_Ls"] = mf.kpts, Ls
        rng = np.random.default_rng(95)
        pts = PeriodicConfigs((rng.random((1, 8, 3)) * 5 - 2) @ sup.lattice_vectors(), sup.lattice_vectors())
        out[f"{tag}_pts"], out[f"{tag}_pts_wrap"] = pts.configs.copy(), pts.wrap.copy()
        for nm, es in (("val", "GTOval_sph"), ("lap", "GTOval_sph_deriv2")):
            out[f"{tag}_mo_{nm}"] = oe.mos(oe.aos(es, pts), 0)
        pbc_protocol(f"{tag}_", sup, {"slater": sl, "jastrow": j2, "wf": wf}, W, 57, electrons, out)
        if tag == "prim":
            N, natm = sum(sup.nelec), sup.natm
            cfg = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(66)).configs.copy() + 20.0, sup.lattice_vectors())
            out["en_configs"], out["en_wrap"] = cfg.configs.copy(), cfg.wrap.copy()
            wf.recompute(cfg)
            with Tapes(730) as t:
                en = pyq.EnergyAccumulator(sup, ewald_gmax=10)(cfg, wf)
            for k, v in en.items():
                out[f"en_{k}"] = np.asarray(v)
            out["en_rot"] = np.asarray(t.log["rot"]).reshape(N, natm, 3, 3)
            out["en_unif"] = np.asarray(t.log["random"]).reshape(N, natm, W)
            nsteps, tstep = 2, 0.5
            start = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(67)).configs.copy(), sup.lattice_vectors())
            out["vmc_start"], out["vmc_start_wrap"] = start.configs.copy(), start.wrap.copy()
            accepts = []
            orig = wf.updateinternals

            def spy(e, epos, c, mask=None, saved_values=None):
                accepts.append(np.asarray(mask).copy())
                return orig(e, epos, c, mask=mask, saved_values=saved_values)

            wf.updateinternals = spy
            with Tapes(803) as t:
                blk, cfg2 = vmc_worker(wf, start, tstep, nsteps, {"energy": pyq.EnergyAccumulator(sup, ewald_gmax=10)})
            wf.updateinternals = orig
            out["vmc_tstep"], out["vmc_nsteps"] = tstep, nsteps
            out["vmc_gauss"] = np.asarray(t.log["normal"]).reshape(nsteps, N, W, 3)
            out["vmc_unif"] = np.asarray(t.log["rand"]).reshape(nsteps, N, W)
            out["vmc_ecp_rot"] = np.asarray(t.log["rot"]).reshape(nsteps, N, natm, 3, 3)
            out["vmc_ecp_unif"] = np.asarray(t.log["random"]).reshape(nsteps, N, natm, W)
            out["vmc_accepts"] = np.asarray(accepts).reshape(nsteps, N, W)
            out["vmc_final"], out["vmc_final_wrap"] = cfg2.configs.copy(), cfg2.wrap.copy()
            out["vmc_final_sign"], out["vmc_final_log"] = wf.value()
            for k, v in blk.items():
                if "time" not in k:
                    out["vmc_blk_" + k] = np.asarray(v)
            print("twisted vmc: wrap moved", np.abs(cfg2.wrap - out["vmc_start_wrap"]).sum())
    save("g20_pbc_twist", **out)


# ------------------------------------------------------------------ G12 DMC propagate + branch
def g_dmc():
    import pyqmc.method.dmc as refdmc
    import pyqmc.wf.orbitals as reforb

    # The un-JIT-ed stand-in for the numba AO evaluator cannot reshape a zero-point result (an all-False
    # T-move mask); PySCF / real numba return an empty array there.  Give the stub the same behaviour.
    orig_aos = reforb.MoleculeOrbitalEvaluator.aos

    def aos(self, eval_str, configs, mask=None):
        coords = configs.configs if mask is None else configs.configs[mask]
        if coords.size == 0:
            nao = self.parameters["mo_coeff_alpha"].shape[0]
            shape = (1, *coords.shape[:-1], nao) if "deriv" not in eval_str else (1, 4 if "deriv1" in eval_str else 5, *coords.shape[:-1], nao)
            return np.zeros(shape)
        return orig_aos(self, eval_str, configs, mask)

    reforb.MoleculeOrbitalEvaluator.aos = aos
    out = {}
    mol = systems.water()
    mf = systems.random_mf(mol)
    wf = make_wf(mol, mf)
    W, nsteps, tstep = 7, 2, 0.02
    configs = walkers(mol, W, 61)
    rng = np.random.default_rng(4)
    configs.configs[:, :2, :] = mol.atom_coords()[0] + 0.3 * rng.standard_normal((W, 2, 3))  # exercise the ECP mask / T-moves
    out["start"] = configs.configs.copy()
    weights = 1.0 + 0.1 * rng.standard_normal(W)
    out["weights0"] = weights.copy()
    e_trial, e_est, branchcut = -17.0, -17.1, 3.0
    out["params"] = np.array([tstep, branchcut, e_trial, e_est, nsteps])
    accepts = []
    orig = wf.updateinternals

    def spy(e, epos, cfg, mask=None, saved_values=None):
        accepts.append(np.asarray(mask).copy())
        return orig(e, epos, cfg, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(900) as t:
        df, configs, weights = refdmc.dmc_propagate(wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps=nsteps,
                                                    accumulators={"energy": pyq.EnergyAccumulator(mol)})
    wf.updateinternals = orig
    out["normal"] = np.asarray(t.log["normal"])
    rand = t.log["rand"]
    out["rand_scalar"] = np.asarray([float(r) for r in rand if np.ndim(r) == 0])
    out["rand_vector"] = np.asarray([r for r in rand if np.ndim(r) == 1])
    out["rand_is_scalar"] = np.asarray([np.ndim(r) == 0 for r in rand])
    out["rot"] = np.asarray(t.log["rot"])
    out["random"] = np.asarray(t.log["random"])
    out["accepts"] = np.asarray(accepts)
    out["final"] = configs.configs.copy()
    out["weights"] = weights
    for k, v in df.items():
        out["df_" + k] = np.asarray(v)
    out["df_keys"] = np.asarray(sorted(df.keys()))
    # branch (dmc.py:342-376)
    w2 = np.abs(1.0 + 0.4 * rng.standard_normal(64))
    cfg2 = OpenConfigs(rng.standard_normal((64, 8, 3)))
    out["branch_weights"], out["branch_configs"] = w2.copy(), cfg2.configs.copy()
    with Tapes(901) as t:
        cfg2, wnew, info = refdmc.branch(cfg2, w2.copy())
    out["branch_u"] = np.asarray(float(t.log["rand"][0]))
    out["branch_newconfigs"], out["branch_newweights"] = cfg2.configs.copy(), wnew
    out["branch_info"] = np.asarray([info["max branches"], info["Number of walkers killed"]])
    save("g12_dmc", **out)


# ------------------------------------------------------------------ G21 parameter-gradient accumulators (SR)
def h2o_multidet_wf():
    """The 12-determinant H2O Slater x two-body x three-body wave function of g_testvalue_many."""
    import pyqmc.wftools as wftools

    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=6)
    dets = systems.random_determinants(mol, mf, 12)
    wf, _ = pyq.generate_wf(mol, mf, jastrow=[pyq.generate_jastrow, wftools.generate_jastrow3], jastrow_kws=[{}, {}],
                            slater_kws=dict(evaluate_orbitals_with="numba", determinants=dets))
    rng = np.random.default_rng(11)
    j2, j3 = wf.wf_factors[1], wf.wf_factors[2]
    j2.parameters["acoeff"] = 0.05 * rng.standard_normal(j2.parameters["acoeff"].shape)
    b = 0.05 * rng.standard_normal(j2.parameters["bcoeff"].shape)
    b[0] = [-0.25, -0.5, -0.25]
    j2.parameters["bcoeff"] = b
    j3.parameters["ccoeff"] = 0.1 * np.random.default_rng(12).standard_normal(j3.parameters["ccoeff"].shape)
    return mol, wf


def sr_to_opt(parameters, seed=21):
    """A partial optimisation mask: first determinant coefficient and the cusp row frozen, random subsets elsewhere."""
    rng = np.random.default_rng(seed)
    to_opt = {}
    for k, v in parameters.items():
        m = rng.random(np.shape(v)) > (0.9 if np.size(v) > 100 else 0.4)
        if k.endswith("det_coeff"):
            m[:] = True
            m[0] = False
        if k.endswith("bcoeff"):
            m[0] = False
        if k.endswith("mo_coeff_beta"):
            m[:] = False  # a key that drops out entirely
        to_opt[k] = m
    return to_opt


def g_sr():
    """LinearTransform (accumulators.py:98-185) and StochasticReconfiguration (stochastic_reconfiguration.py:49-176) on the
    multi-determinant H2O wave function.  The energy accumulator is replaced by recorded arrays (the ECP energy draws a
    random quadrature rotation), with two walkers inside the nodal cut-off."""
    from pyqmc.observables.accumulators import LinearTransform
    from pyqmc.observables.stochastic_reconfiguration import StochasticReconfiguration

    mol, wf = h2o_multidet_wf()
    W = 6
    configs = walkers(mol, W, 81)
    wf.recompute(configs)
    rng = np.random.default_rng(210)
    en = {"total": -17.0 + rng.standard_normal(W), "grad2": np.abs(rng.standard_normal(W)) * 40 + 5.0, "ke": rng.standard_normal(W)}
    en["grad2"][1], en["grad2"][4] = 3.0e6, 2.5e7

    class Recorded:
        def __call__(self, configs, wf):
            return {k: v.copy() for k, v in en.items()}

        def keys(self):
            return set(en)

        def shapes(self):
            return {k: () for k in en}

    to_opt = sr_to_opt(wf.parameters)
    tr = LinearTransform(wf.parameters, to_opt)
    sr = StochasticReconfiguration(Recorded(), tr, nodal_cutoff=1e-3, eps=1e-2)
    out = {"configs": configs.configs.copy(), "en_keys": np.asarray(sorted(en)), "opt_keys": np.asarray(list(to_opt.keys()))}
    for k, m in to_opt.items():
        out["opt_" + k] = m
    for k, v in en.items():
        out["en_" + k] = v
    out["nparams"] = np.asarray(tr.nparams)
    out["ser_params"] = tr.serialize_parameters(wf.parameters)
    out["ser_grads"] = np.asarray(tr.serialize_gradients(wf.pgradient()))
    d = sr(configs, wf)
    for k in ("dpH", "dppsi", "dpidpj"):
        out["call_" + k] = np.asarray(d[k])
    weights = np.abs(1.0 + 0.3 * rng.standard_normal(W))
    out["weights"] = weights
    for tag, wts in (("avg", None), ("wavg", weights)):
        d = sr.avg(configs, wf, weights=wts)
        out[tag + "_keys"] = np.asarray(sorted(d))
        for k, v in d.items():
            out[tag + "_" + k] = np.asarray(v)
    data = sr.avg(configs, wf, weights=weights)
    for strat in ("pseudo_inverse", "regularized_inverse"):
        sr.inverse_strategy = strat
        dp, report = sr.delta_p([0.1, 0.25], data)
        out["dp_" + strat] = np.asarray(dp)
        out["report_" + strat] = np.asarray([report["pgrad"], report["SRdot"]])
    x = tr.serialize_parameters(wf.parameters) + 0.01 * rng.standard_normal(tr.nparams)
    out["deser_x"] = x
    new = tr.deserialize(wf, x)
    out["deser_keys"] = np.asarray(sorted(new))
    for k, v in new.items():
        out["deser_" + k] = np.asarray(v)
    save("g21_sr", **out)


# ------------------------------------------------------------------ G22 one-body density matrix accumulator
def g_obdm():
    """OBDMAccumulator (obdm.py:26-213) on the H2O Slater-Jastrow wave function and on the 12-determinant one, with
    numpy.random seeded: two successive evaluations (the auxiliary walkers carry over) and an average."""
    import pyqmc.observables.obdm as refobdm
    from pyqmc.wf.orbitals import MoleculeOrbitalEvaluator

    out = {}
    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=6)
    dets = systems.random_determinants(mol, mf, 12)
    cases = {"sj": make_wf(mol, systems.random_mf(mol)), "md": make_wf(mol, mf, determinants=dets)}
    orb = np.asarray(mf.mo_coeff)[:, :7] if np.ndim(mf.mo_coeff) == 2 else np.asarray(mf.mo_coeff[0])[:, :7]
    out["orb_coeff"] = orb
    W = 7
    for tag, wf in cases.items():
        configs = walkers(mol, W, 61)
        out[tag + "_configs"] = configs.configs.copy()
        wf.recompute(configs)
        for sub, kw in (("up", dict(spin=0)), ("some", dict(electrons=np.array([1, 5, 6]), naux=10)), ("all", dict())):
            acc = refobdm.OBDMAccumulator(mol, orb, nsweeps=3, tstep=0.4, warmup=6, **kw)
            acc.orbitals = MoleculeOrbitalEvaluator(mol, [orb, orb], evaluate_orbitals_with="numba")
            np.random.seed(220 + len(out))
            out[f"{tag}_{sub}_seed"] = np.asarray(220 + len(out))
            for call in (0, 1):
                d = acc(configs, wf)
                out[f"{tag}_{sub}_value{call}"], out[f"{tag}_{sub}_norm{call}"] = d["value"], d["norm"]
            d = acc.avg(configs, wf)
            out[f"{tag}_{sub}_avg_value"], out[f"{tag}_{sub}_avg_norm"] = d["value"], d["norm"]
            out[f"{tag}_{sub}_aux_final"] = acc._extra_config.configs.copy()
    out["normalized"] = refobdm.normalize_obdm(out["sj_all_avg_value"], out["sj_all_avg_norm"])
    save("g22_obdm", **out)


# ------------------------------------------------------------------ G23 two-body density matrix accumulator
def g_tbdm():
    """TBDMAccumulator (tbdm.py:63-283) on the H2O Slater-Jastrow wave function, sectors (0,1) and (0,0), numpy.random seeded."""
    import pyqmc.observables.tbdm as reftbdm
    from pyqmc.wf.orbitals import MoleculeOrbitalEvaluator

    out = {}
    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=3)
    wf = make_wf(mol, systems.random_mf(mol))
    C = np.asarray(mf.mo_coeff)
    orb = [C[0][:, :5], C[1][:, 1:5]]  # different bases (and sizes) for the two spins
    out["orb_up"], out["orb_dn"] = orb[0], orb[1]
    W = 5
    configs = walkers(mol, W, 71)
    out["configs"] = configs.configs.copy()
    wf.recompute(configs)
    rng = np.random.default_rng(230)
    some = rng.integers(0, 4, size=(30, 4))
    out["ijkl_some"] = some
    for tag, kw in (("ud", dict(spin=(0, 1))), ("uu", dict(spin=(0, 0), ijkl=some, naux=9)), ("du", dict(spin=(1, 0), ijkl=some))):
        acc = reftbdm.TBDMAccumulator(mol, orb, nsweeps=2, tstep=0.4, warmup=4, **kw)
        acc.orbitals = MoleculeOrbitalEvaluator(mol, orb, evaluate_orbitals_with="numba")
        np.random.seed(231 + len(out))
        out[tag + "_seed"] = np.asarray(231 + len(out))
        for call in (0, 1):
            d = acc(configs, wf)
            for k in ("value", "norm_a", "norm_b"):
                out[f"{tag}_{k}{call}"] = d[k]
        d = acc.avg(configs, wf)
        for k in ("value", "norm_a", "norm_b"):
            out[f"{tag}_avg_{k}"] = d[k]
        out[tag + "_aux_final_a"], out[tag + "_aux_final_b"] = acc._aux_configs[0].configs.copy(), acc._aux_configs[1].configs.copy()
        out[tag + "_final_log"] = wf.value()[1]  # the state after all the there-and-back updates
    nmo = (5, 4)
    full = out["ud_avg_value"].reshape(nmo[0], nmo[0], nmo[1], nmo[1])
    out["normalized"] = reftbdm.normalize_tbdm(full, out["ud_avg_norm_a"], out["ud_avg_norm_b"])
    save("g23_tbdm", **out)


# ------------------------------------------------------------------ G24 parameter gradients of a periodic Slater-Jastrow
def g_pbc_pgrad():
    """wf.pgradient() (slater.py:462-542 with PBCOrbitalEvaluatorKpoints.pgradient orbitals.py:239-254) on the diamond
    cells with real Bloch phases, plus the parameter arrays themselves: the orbital coefficients are per-k blocks
    (nao_prim, nmo_k) concatenated over k."""
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    for tag, W in (("gamma", 4), ("fcc2cubic", 3)):
        sup, wf = ref_pbc_wf(tag)
        cfg = PeriodicConfigs(systems.initial_guess(sup, W, rng=np.random.default_rng(240)).configs.copy(), sup.lattice_vectors())
        out[tag + "_configs"] = cfg.configs.copy()
        wf.recompute(cfg)
        pg = wf.pgradient()
        out[tag + "_keys"] = np.asarray(sorted(pg.keys()))
        for k, v in pg.items():
            out[f"{tag}_pgrad_{k}"] = np.asarray(v)
        for k, v in wf.parameters.items():
            out[f"{tag}_param_{k}"] = np.asarray(v)
    save("g24_pbc_pgrad", **out)


# ------------------------------------------------------------------ G25 testvalue_many with complex determinants
def g_complex_testvalue_many():
    """testvalue_many (slater.py:448-460, multiplywf.py:112-114) for complex Bloch orbitals at zero twist and for a
    twisted cell, auxiliary positions inside and outside the cell (wrap phase of the moved electron)."""
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    sup, mf, Ls, oe, sl, j2, wf = ref_pbc_wf_complex()
    cases = {"cplx": (sup, sl, j2, wf, 3, np.array([0, 5, 6, 11]))}
    sup2, mf2, Ls2, oe2, sl2, j22, wf2, W2, el2 = ref_twisted_wf("s211")
    cases["twist"] = (sup2, sl2, j22, wf2, 3, np.array([1, 7, 8, 15]))
    for tag, (cell, slater, jas, full, W, es) in cases.items():
        rng = np.random.default_rng(250 + len(out))
        cfg = PeriodicConfigs(systems.initial_guess(cell, W, rng=rng).configs.copy(), cell.lattice_vectors())
        out[tag + "_configs"], out[tag + "_wrap"] = cfg.configs.copy(), cfg.wrap.copy()
        full.recompute(cfg)
        aux = (rng.random((W, 3)) * 4 - 1.5) @ cell.lattice_vectors()
        out[tag + "_aux"], out[tag + "_es"] = aux, es
        epos = cfg.make_irreducible(0, aux)
        for nm, w in (("slater", slater), ("j2", jas), ("wf", full)):
            out[f"{tag}_{nm}"] = np.asarray(w.testvalue_many(es, epos))
    save("g25_complex_testvalue_many", **out)


# ------------------------------------------------------------------ G31 pgradient of complex determinants
def g_complex_pgrad():
    """wf.pgradient() (slater.py:462-542, PBCOrbitalEvaluatorKpoints.pgradient orbitals.py:239-254) for complex Bloch orbitals:
    the 3x1x1 supercell at k = 0, 1/3, 2/3 b1 (complex phases e^{ik.L}, complex coefficients) and the twisted cells of g20, walkers
    partly OUTSIDE the cell (wrap counters != 0: the derivative carries the wrap phase of every electron).  The orbital parameters
    are the per-k blocks (nao_prim, nmo_k) concatenated over k; their derivatives are complex and holomorphic (no conjugation)."""
    from pyqmc.configurations.coord import PeriodicConfigs

    out = {}
    sup, mf, Ls, oe, sl, j2, wf = ref_pbc_wf_complex()
    cases = {"cplx": (sup, sl, wf, 3)}
    for tag in TWIST_CASES:
        sup2, mf2, Ls2, oe2, sl2, j22, wf2, W2, el2 = ref_twisted_wf(tag)
        cases["twist_" + tag] = (sup2, sl2, wf2, 3)
    for tag, (cell, slater, full, W) in cases.items():
        rng = np.random.default_rng(310 + len(out))
        true_x = systems.initial_guess(cell, W, rng=rng).configs + (rng.integers(-1, 2, size=(W, sum(cell.nelec), 3)) @ cell.lattice_vectors())
        cfg = PeriodicConfigs(true_x.copy(), cell.lattice_vectors())  # folds the walkers and sets their wrap counters
        out[tag + "_configs"], out[tag + "_wrap"] = cfg.configs.copy(), cfg.wrap.copy()
        assert np.abs(cfg.wrap).max() > 0
        full.recompute(cfg)
        for nm, w in (("slater", slater), ("wf", full)):
            pg = w.pgradient()
            out[f"{tag}_{nm}_keys"] = np.asarray(sorted(pg.keys()))
            for k, v in pg.items():
                out[f"{tag}_{nm}_pgrad_{k}"] = np.asarray(v)
        for k, v in slater.parameters.items():
            out[f"{tag}_param_{k}"] = np.asarray(v)
    save("g31_complex_pgrad", **out)


# ------------------------------------------------------------------ G32 periodic g / h shells
def g_pbc_high_l():
    """Lattice-summed shells with l = 4, 5 (pbcgto.py:52-96: SPH4 / SPH5 wrappers): the diamond primitive cell with a g and an h
    shell on carbon, as a Gamma cell and as the fcc -> cubic 4-fold supercell (4 k-points, real phases): AOs and MOs (value,
    gradient, Laplacian) at points inside and outside the cell, and the Slater / Jastrow / product protocol with moves."""
    import pyqmc.wftools as wftools
    from pyqmc.configurations.coord import PeriodicConfigs
    from pyqmc.wf.multiplywf import MultiplyWF
    from pyqmc_amd import pbc as mypbc

    prim = systems.diamond_primitive_high_l()
    out = {}
    for tag in ("gamma", "fcc2cubic"):
        S, W, electrons = PBC_SLATER_CASES[tag]
        sup = mypbc.get_supercell(prim, S)
        mf = mypbc.random_kmf(sup)
        Ls = mypbc.lattice_points_within(prim.lattice_vectors(), 30.0)
        ev, oe, sl = ref_pbc_objects(sup, mf.kpts, mf.mo_coeff, Ls)
        assert max(ev.max_l) == 5
        rng = np.random.default_rng(321)
        pts = PeriodicConfigs((rng.random((1, 10, 3)) * 3 - 1) @ sup.lattice_vectors(), sup.lattice_vectors())
        out[f"{tag}_pts"], out[f"{tag}_pts_wrap"] = pts.configs.copy(), pts.wrap.copy()
        for nm, es in (("val", "GTOval_sph"), ("grad", "GTOval_sph_deriv1"), ("lap", "GTOval_sph_deriv2")):
            ao = oe.aos(es, pts)
            out[f"{tag}_ao_{nm}"] = ao
            out[f"{tag}_mo_{nm}"] = oe.mos(ao, 0)
        j2, _ = wftools.generate_jastrow(sup)
        jr = np.random.default_rng(17)
        j2.parameters["acoeff"] = 0.05 * jr.standard_normal(j2.parameters["acoeff"].shape)
        b = 0.05 * jr.standard_normal(j2.parameters["bcoeff"].shape)
        b[0] = [-0.25, -0.5, -0.25]
        j2.parameters["bcoeff"] = b
        wf = MultiplyWF(sl, j2)
        pbc_protocol(f"{tag}_", sup, {"slater": sl, "jastrow": j2, "wf": wf}, W, 51, electrons, out)
    save("g32_pbc_high_l", **out)


# ------------------------------------------------------------------ G27 on-disk layout (hdftools)
class _FakeDataset:
    def __init__(self, shape, dtype):
        self.a = np.zeros(shape, dtype=dtype if dtype is not None else float)

    shape = property(lambda self: self.a.shape)
    dtype = property(lambda self: self.a.dtype)

    def resize(self, shape, axis=None):
        shape = tuple(shape) if axis is None else self.a.shape[:axis] + (shape,) + self.a.shape[axis + 1:]
        new = np.zeros(shape, dtype=self.a.dtype)
        sl = tuple(slice(0, min(o, n)) for o, n in zip(self.a.shape, shape))
        new[sl] = self.a[sl]
        self.a = new

    def __setitem__(self, k, v):
        self.a[k] = v

    def __getitem__(self, k):
        return self.a[k]


class _FakeH5File(dict):
    """In-memory stand-in for h5py.File: records what the reference creates (names, shapes, dtypes, attributes)."""
    store = {}

    def __new__(cls, path, mode="r"):
        if path not in cls.store:
            cls.store[path] = dict.__new__(cls)
            cls.store[path].attrs = {}
        return cls.store[path]

    def __init__(self, path, mode="r"):
        pass

    def create_dataset(self, name, shape=None, maxshape=None, dtype=None, chunks=None, data=None):
        self[name] = _FakeDataset(shape, dtype)
        return self[name]

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def g_hdf_layout():
    """Names, shapes and dtype kinds of everything the reference's vmc (mc.py:92-99,262-265) and rundmc (dmc.py:379-391)
    put into their HDF5 file, recorded through an in-memory h5py stand-in (no HDF5 library in this image)."""
    import json

    import pyqmc.method.dmc as refdmc
    import pyqmc.method.mc as refmc

    import pyqmc.wf.orbitals as reforb

    orig_aos = reforb.MoleculeOrbitalEvaluator.aos

    def aos(self, eval_str, configs, mask=None):  # the un-JIT-ed AO stand-in cannot reshape a zero-point result (see g_dmc)
        coords = configs.configs if mask is None else configs.configs[mask]
        if coords.size == 0:
            nao = self.parameters["mo_coeff_alpha"].shape[0]
            return np.zeros((1, *coords.shape[:-1], nao) if "deriv" not in eval_str else (1, 4 if "deriv1" in eval_str else 5, *coords.shape[:-1], nao))
        return orig_aos(self, eval_str, configs, mask)

    reforb.MoleculeOrbitalEvaluator.aos = aos
    fake = types.SimpleNamespace(File=_FakeH5File)
    refmc.h5py = refdmc.h5py = fake
    mol = systems.water()
    mf = systems.random_mf(mol)
    out = {}
    np.random.seed(3)
    wf = make_wf(mol, mf)
    cfg = walkers(mol, 6, 1)
    refmc.vmc(wf, cfg, nblocks=3, nsteps_per_block=2, tstep=0.3, accumulators={"energy": pyq.EnergyAccumulator(mol)}, hdf_file="vmc.h5")
    f = _FakeH5File.store["vmc.h5"]
    out["vmc"] = {k: [list(v.shape), v.dtype.kind] for k, v in f.items()}
    out["vmc_attrs"] = sorted(f.attrs)
    refdmc.rundmc(wf, walkers(mol, 6, 2), tstep=0.05, nblocks=2, nsteps_per_block=2, vmc_warmup=1,
                  accumulators={"energy": pyq.EnergyAccumulator(mol)}, hdf_file="dmc.h5")
    f = _FakeH5File.store["dmc.h5"]
    out["dmc"] = {k: [list(v.shape), v.dtype.kind] for k, v in f.items()}
    out["dmc_attrs"] = sorted(f.attrs)
    save("g27_hdf_layout", layout=np.array(json.dumps(out, sort_keys=True)))


def g_chk_mol():
    """SURVEY 8(f4), second half: the `mol` JSON of the reference's own PySCF checkpoint files (tests/files/*.hdf5, copied as data
    to tests/golden/files/) -> what the REFERENCE builds from it.  The JSON is pulled out here with an independent regular
    expression (no HDF5 library in the image, and not pyqmc_amd.chkfile, which is the thing under test); a duck-typed molecule
    with the file's _basis / _ecp / _atom goes through the reference's AtomicOrbitalEvaluator (normalised shell tables,
    numba/gto.py:435-470), its AO evaluator on random points, and its ECP functors (eval_ecp.py:160-200) on a radial grid.
    Lattice vectors in bohr are `a` / pyscf's BOHR (Cell.lattice_vectors(); pyscf itself cannot run here: stated, not measured)."""
    import json
    import re

    out = {}
    for name in ("diamond_primitive", "li_cubic_ccecp"):
        raw = open(os.path.join("/root/reference/tests/files", name + ".hdf5"), "rb").read()
        m = re.search(rb'\{"atom": .*?"precision": [0-9.e+-]+\}', raw, re.S)
        d = json.loads(m.group(0).decode())
        syms = [a[0] for a in d["_atom"]]
        xyz = np.array([a[1] for a in d["_atom"]])
        ref_charges = [row[0] for row in d["_atm"]]

        class M:
            _basis, _ecp, _atom, natm, cart = d["_basis"], d["_ecp"], d["_atom"], len(syms), False

            def atom_coords(self):
                return xyz

            def atom_pure_symbol(self, i):
                return syms[i]

            def atom_symbol(self, i):
                return syms[i]

            def atom_charges(self):
                return np.array(ref_charges, dtype=float)

        mol = M()
        ev = refgto.AtomicOrbitalEvaluator(mol)
        pts = np.random.default_rng(28).standard_normal((20, 3)) * 2.0 + xyz[0]
        out[name + "_basis_ls"] = np.asarray(ev.basis_ls)
        out[name + "_basis_arrays"] = np.asarray(ev.basis_arrays)
        out[name + "_splits"] = np.asarray(ev.splits)
        out[name + "_pts"] = pts
        out[name + "_ao"] = np.asarray(ev.eval_gto("GTOval_sph", pts))
        out[name + "_ao_lap"] = np.asarray(ev.eval_gto("GTOval_sph_deriv2", pts))
        r = np.linspace(0.05, 4.0, 60)
        for sym in d["_ecp"]:
            ls, vl = eval_ecp.get_v_l(mol, sym, r)
            out[f"{name}_ecp_{sym}_l"] = np.array(list(ls))
            out[f"{name}_ecp_{sym}_v"] = vl
        out[name + "_ecp_r"] = r
        out[name + "_xyz"] = xyz
        out[name + "_charges"] = np.array(ref_charges, dtype=float)
        bohr_input = str(d.get("unit", "angstrom")).lower().startswith(("b", "au"))  # Cell.lattice_vectors(): a / BOHR unless the input unit was bohr
        out[name + "_lattice_bohr"] = np.array(d["a"], dtype=float) / (1.0 if bohr_input else 0.52917721092)
        out[name + "_syms"] = np.array(syms)
    save("g28_chk_mol", **out)


def g_pbc_complex_dmc():
    """dmc_propagate on a COMPLEX periodic wave function (3x1x1 diamond supercell, k = 0, 1/3, 2/3 b1, complex coefficients) with
    ECP T-moves.  The reference's propose_tmoves compares complex amplitudes with `> 0` / `< 0` and stores 1 / ratio into a real
    array (dmc.py:83-101): under NumPy >= 2 the comparisons raise, under NumPy 1 they ordered by the real part first and the
    assignment dropped the imaginary part.  The rule pinned here is the one that keeps every line of propose_tmoves as it is and
    is what fixed-phase practice uses: T-move amplitudes are formed from Re[Psi(R')/Psi(R)] — realised by a shim around
    EnergyAccumulator.nonlocal_tmoves that hands the reference the real part of the ratios.  Everything else (complex
    drift-diffusion without a node constraint, complex ECP energies, weights from Re E_L) is the reference untouched."""
    import pyqmc.method.dmc as refdmc
    import pyqmc.wf.orbitals as reforb
    from pyqmc.configurations.coord import PeriodicConfigs
    from pyqmc.observables.accumulators import EnergyAccumulator as RefAcc

    orig_aos = reforb.PBCOrbitalEvaluatorKpoints.aos

    def aos(self, eval_str, configs, mask=None):  # the un-JIT-ed AO path cannot take an empty point list (see g_pbc_dmc)
        coords = configs.configs if mask is None else configs.configs[mask]
        if coords.size == 0:
            nao, nk = self.parameters["mo_coeff_alpha"].shape[0], len(self._kpts)
            comp = () if "deriv" not in eval_str else ((4,) if "deriv1" in eval_str else (5,))
            return np.zeros((nk, *comp, *coords.shape[:-1], nao))
        return orig_aos(self, eval_str, configs, mask)

    reforb.PBCOrbitalEvaluatorKpoints.aos = aos
    orig_tm = RefAcc.nonlocal_tmoves

    def real_part_tmoves(self, configs, wf, e, tau):
        m = orig_tm(self, configs, wf, e, tau)
        m["ratio"] = np.real(m["ratio"])
        return m

    RefAcc.nonlocal_tmoves = real_part_tmoves
    out = {}
    sup, mf, Ls, oe, sl, j2, wf = ref_pbc_wf_complex()
    assert sl.dtype == complex
    W, nsteps, tstep = 4, 2, 0.1
    rng = np.random.default_rng(6)
    N = sum(sup.nelec)
    base = systems.initial_guess(sup, W, rng=np.random.default_rng(66)).configs.copy()
    # pull a few electrons close to carbon cores so the ECP mask passes and T-moves have weight
    for k, e in enumerate((0, 3, 7, 12, 15, 20)):
        base[:, e] = sup.atom_coords()[k % sup.natm][None] + 0.25 * rng.standard_normal((W, 3))
    configs = PeriodicConfigs(base, sup.lattice_vectors())
    out["start"], out["start_wrap"] = configs.configs.copy(), configs.wrap.copy()
    weights = 1.0 + 0.1 * rng.standard_normal(W)
    out["weights0"] = weights.copy()
    e_trial, e_est, branchcut = -30.0, -30.5, 5.0
    out["params"] = np.array([tstep, branchcut, e_trial, e_est, nsteps])
    accepts = []
    orig = wf.updateinternals

    def spy(e, epos, cfg, mask=None, saved_values=None):
        accepts.append(np.asarray(mask).copy())
        return orig(e, epos, cfg, mask=mask, saved_values=saved_values)

    wf.updateinternals = spy
    with Tapes(951) as t:
        df, configs, weights = refdmc.dmc_propagate(wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps=nsteps,
                                                    accumulators={"energy": pyq.EnergyAccumulator(sup, ewald_gmax=10)})
    wf.updateinternals = orig
    reforb.PBCOrbitalEvaluatorKpoints.aos = orig_aos
    RefAcc.nonlocal_tmoves = orig_tm
    out["normal"] = np.asarray(t.log["normal"])
    rand = t.log["rand"]
    out["rand_scalar"] = np.asarray([float(r) for r in rand if np.ndim(r) == 0])
    out["rand_vector"] = np.asarray([r for r in rand if np.ndim(r) == 1])
    out["rot"] = np.asarray(t.log["rot"])
    out["random"] = np.asarray(t.log["random"])
    out["accepts"] = np.asarray(accepts)
    out["final"], out["final_wrap"] = configs.configs.copy(), configs.wrap.copy()
    out["weights"] = weights
    for k, v in df.items():
        out["df_" + k] = np.asarray(v)
    out["df_keys"] = np.asarray(sorted(df.keys()))
    nt = len(accepts) // 2  # first half of every step: the T-move updates
    print("complex pbc dmc: tmove acc", df.get("tmove_acceptance"), "acc", df.get("acceptance"), "E", df.get("energytotal"),
          "T-moves accepted", int(np.sum([a.sum() for a in accepts[:N]])), "+", int(np.sum([a.sum() for a in accepts[2 * N : 3 * N]])))
    save("g30_pbc_complex_dmc", **out)


if __name__ == "__main__":
    if len(sys.argv) > 1:  # regenerate only the named fixtures: python make_golden.py g_sr g_obdm
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    g_sherman_morrison()
    g_ao()
    g_ao_high_l()
    g_func3d()
    g_protocol()
    g_energy()
    g_vmc()
    g_jastrow3()
    g_dmc()
    g_pbc()
    g_pbc_slater()
    g_pbc_energy()
    g_pbc_dmc()
    g_testvalue_many()
    g_pbc_complex()
    g_pbc_twist()
    g_sr()
    g_obdm()
    g_tbdm()
    g_pbc_pgrad()
    g_complex_testvalue_many()
    g_hdf_layout()
    g_chk_mol()
    g_pbc_complex_dmc()
    g_complex_pgrad()
    g_pbc_high_l()
    g_ecp_naip()
    g_ecp_batched()
    g_big()
    g_vmc_cluster()
    g_ao_general()
