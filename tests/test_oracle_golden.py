"""Pin the CPU oracle against golden vectors produced by the real reference
(tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

import helpers
from helpers import golden, relerr
from oracle import energy as oenergy
from oracle import gto, jastrow_basis, vmc as ovmc
from pyqmc_amd import systems
from pyqmc_amd.configs import OpenConfigs


def test_g1_sherman_morrison():
    """Reference recipe tests/unit/test_sherman_morrison.py:32-82 (1e-13)."""
    g = golden("g1_sherman_morrison")
    for tag in ("small", "n32"):
        inv, vec, e = g[tag + "_inv"], g[tag + "_vec"], int(g[tag + "_e"])
        tmp = np.einsum("wdk,wdkj->wdj", vec, inv)
        ratio = tmp[:, :, e]
        invr = inv[:, :, :, e] / ratio[:, :, None]
        new = inv - np.einsum("wdi,wdj->wdij", invr, tmp)
        new[:, :, :, e] = invr
        assert np.max(np.abs(ratio - g[tag + "_ratio"])) < 1e-13
        assert np.max(np.abs(new - g[tag + "_invnew"])) < 1e-12


@pytest.mark.parametrize("tag,mol", [("h2o", systems.water()), ("c2", systems.carbon_dimer())])
def test_g2_ao(tag, mol):
    g = golden("g2_ao")
    table = gto.AOTable(mol)
    pts = g[tag + "_pts"]
    assert relerr(gto.eval_ao(table, pts, 1)[0], g[tag + "_val"]) < 1e-13
    assert relerr(gto.eval_ao(table, pts, 4), g[tag + "_deriv1"]) < 1e-13
    assert relerr(gto.eval_ao(table, pts, 5), g[tag + "_deriv2"]) < 1e-12


def test_g38_general_contractions():
    """A generally contracted basis (several coefficient columns over one set of exponents: every all-electron cc-pVXZ set, and what the
    reference's default AO path ``mol.eval_gto`` takes, orbitals.py:46-51) is split into single-column shells in PySCF's AO order
    (tables.split_general_contractions).  (a) The oracle on the split shells against the reference's in-repo evaluator fed the same
    shells (golden g38).  (b) libcgto-free check that the split is exact: the s and p functions evaluated straight from the generally
    contracted tables — one exponential per primitive, the columns of the coefficient matrix normalised as PySCF does — equal them."""
    from pyqmc_amd import tables

    g = golden("g38_ao_general")
    mol = systems.water_general()
    table = gto.AOTable(mol)
    for nc, key in ((1, "val"), (4, "deriv1"), (5, "deriv2")):
        out = gto.eval_ao(table, g["pts"], nc)
        assert relerr(out[0] if nc == 1 else out, g[key]) < 1e-13, key
    # (b)
    import math

    val = gto.eval_ao(table, g["pts"], 1)[0]
    col = 0
    for ia in range(mol.natm):
        raw = {"O": systems._O_GENERAL, "H": systems._H_GENERAL}[mol.atom_pure_symbol(ia)]
        d = g["pts"] - mol.atom_coords()[ia]
        r2 = np.sum(d * d, axis=1)
        for sh in raw:
            l, rows = sh[0], np.asarray(sh[1:], dtype=float)
            e, C = rows[:, 0], rows[:, 1:]
            m = l + 1.5
            C = C * np.sqrt(2.0 * (2.0 * e) ** m / math.gamma(m))[:, None]  # gto_norm of every primitive
            S = math.gamma(m) / (2.0 * (e[:, None] + e[None, :]) ** m)
            C = C / np.sqrt(np.einsum("pi,pq,qi->i", C, S, C))[None, :]  # every column normalised on its own
            R = np.exp(-np.outer(r2, e)) @ C  # (points, columns)
            for k in range(C.shape[1]):
                if l == 0:
                    assert relerr(val[:, col], 0.28209479177387814 * R[:, k]) < 1e-13
                elif l == 1:
                    assert relerr(val[:, col:col + 3], 0.4886025119029199 * d * R[:, k, None]) < 1e-13
                col += 2 * l + 1
    assert col == mol.nao() == 24
    # the splitter itself: kappa = 0 entries are accepted, other kappas and ragged rows refused
    assert tables.split_general_contractions([[1, 0, [2.0, 1.0, 0.0], [0.5, 0.3, 1.0]]]) == [[1, [2.0, 1.0], [0.5, 0.3]], [1, [0.5, 1.0]]]
    with pytest.raises(NotImplementedError):
        tables.split_general_contractions([[1, -2, [2.0, 1.0]]])
    with pytest.raises(ValueError):
        tables.split_general_contractions([[0, [2.0, 1.0, 0.5], [1.0, 1.0]]])


def test_g26_ao_high_l():
    """f, g, h shells (numba/gto.py:107-118 supports l <= 5): oracle (l <= 3 written out, l = 4, 5 from the generated
    monomial tables) against the reference's evaluator."""
    g = golden("g26_ao_high_l")
    table = gto.AOTable(systems.carbon_dimer_high_l())
    assert table.max_l == int(np.max(g["max_l"])) == 5
    for nc, key in ((1, "val"), (4, "deriv1"), (5, "deriv2")):
        out = gto.eval_ao(table, g["pts"], nc)
        assert relerr(out[0] if nc == 1 else out, g[key]) < 1e-13, key


def test_g4_func3d():
    g = golden("g4_func3d")
    r, rvec = g["r"], g["rvec"]
    cases = {"pade_2.0_1.5": ([("pade", 2.0)], 1.5), "cusp_2.0_1.5": ([("cusp", 2.0)], 1.5),
             "pade_0.2_7.5": ([("pade", 0.2)], 7.5), "cusp_24_7.5": ([("cusp", 24.0)], 7.5)}
    for k, (basis, rcut) in cases.items():
        inside = r < rcut  # the bare functions are only meaningful inside; evaluator zeroes outside
        gr, v = jastrow_basis.evaluate(basis, rcut, rvec, r, "gradient_value")
        assert np.max(np.abs(v[inside, 0] - g[k + "_value"][inside])) < 1e-14
        assert np.max(np.abs(gr[inside, 0] - g[k + "_grad"][inside])) < 1e-13
        gr, l = jastrow_basis.evaluate(basis, rcut, rvec, r, "gradient_laplacian")
        ok = inside & np.isfinite(g[k + "_lap"])
        assert np.max(np.abs(l[ok, 0] - g[k + "_lap"][ok])) < 1e-12
    basis, rcut = [("cusp", 24.0), ("pade", 0.5)], 1.5
    assert np.max(np.abs(jastrow_basis.evaluate(basis, rcut, rvec, r, "value") - g["eval_value"])) < 1e-14
    gr, v = jastrow_basis.evaluate(basis, rcut, rvec, r, "gradient_value")
    assert np.max(np.abs(gr - g["eval_grad"])) < 1e-13 and np.max(np.abs(v - g["eval_gv_value"])) < 1e-14
    gr, l = jastrow_basis.evaluate(basis, rcut, rvec, r, "gradient_laplacian")
    assert np.max(np.abs(l - g["eval_lap"][..., 0])) < 1e-12
    assert np.all(v[r >= rcut] == 0) and np.all(l[r >= rcut] == 0)


def test_default_basis_matches_reference_parameters():
    """wftools.py:64-96 values quoted in SURVEY 8(a) a9."""
    ab, bb, rcut = jastrow_basis.default_basis()
    assert rcut == 7.5 and bb[0] == ("cusp", 24.0)
    assert np.allclose([b for _, b in ab], [0.2, 4.9437, 28.439, 144.81], rtol=2e-4)
    assert np.allclose([b for _, b in bb[1:]], [0.5, 6.4296, 35.799], rtol=2e-4)


@pytest.mark.parametrize("name", ["g5_protocol_h2o", "g8_protocol_h2o_multidet", "g5_protocol_cluster"])
def test_g5_g8_protocol(name):
    mol, mf, dets, g = helpers.case(name)
    wf = helpers.oracle_wf(mol, mf, dets)
    err = helpers.run_protocol(wf, g)
    # the cluster fixture force-accepts a move with |ratio| ~ 9e-5 (near-singular Slater matrix
    # afterwards, cond ~ 1e7), which amplifies summation-order roundoff; see DESIGN.md "tolerances"
    tol = 5e-9 if name == "g5_protocol_cluster" else 1e-10
    bad = {k: v for k, v in err.items() if v > tol}
    assert not bad, bad
    # internals
    configs = OpenConfigs(g["configs"].copy())
    wf.recompute(configs)
    sl, ja = wf.wf_factors
    for s in (0, 1):
        assert relerr(sl._inverse[s], g[f"slater_inverse{s}"]) < 1e-10
        assert relerr(sl._dets[s], g[f"slater_dets{s}"]) < 1e-11
    assert relerr(ja._avalues, g["jastrow_avalues"]) < 1e-12
    assert relerr(ja._bvalues, g["jastrow_bvalues"]) < 1e-12


@pytest.mark.parametrize("tag,mol,W", [("h2o", systems.water(), 8), ("cluster", systems.water_cluster(), 2)])
def test_g10_energy(tag, mol, W):
    g = golden("g10_energy")
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g[tag + "_configs"].copy())
    wf.recompute(configs)
    for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
        en = oenergy.energy(mol, configs, wf, thr, g[f"{tag}_{thr_tag}_rot"], g[f"{tag}_{thr_tag}_unif"])
        for k in ("ke", "ee", "ei", "ecp", "grad2", "total"):
            assert relerr(en[k], g[f"{tag}_{thr_tag}_{k}"]) < 1e-9, (thr_tag, k)
    assert abs(oenergy.coulomb(mol, configs)[2] - float(np.ravel(g[tag + "_ii"])[0])) < 1e-10


def test_g33_ecp_quadrature_rules():
    """eval_ecp.ecp(..., naip) (eval_ecp.py:21-40, :228-252) for every grid the reference tabulates (:278-336), on an oxygen with
    s, p, d non-local channels; the grids themselves bit for bit."""
    g = golden("g33_ecp_naip")
    for naip in (6, 12, 18, 26, 32, 50):
        pts, wts = oenergy.quadrature(naip)
        assert np.array_equal(pts, g[f"grid{naip}_points"]) and np.array_equal(wts, g[f"grid{naip}_weights"])
    with pytest.raises(ValueError):
        oenergy.quadrature(14)
    mol = systems.water_multichannel()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g["configs"].copy())
    wf.recompute(configs)
    for naip in (None, 6, 18, 26, 32, 50):
        for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
            tag = f"naip{naip}_{thr_tag}"
            en = oenergy.energy(mol, configs, wf, thr, g[tag + "_rot"], g[tag + "_unif"], naip=naip)
            assert relerr(en["ecp"], g[tag + "_ecp"]) < 1e-10 and relerr(en["total"], g[tag + "_total"]) < 1e-10, tag
    assert relerr(g["naip50_det_ecp"], g["naip18_det_ecp"]) > 1e-6  # the rules really differ on this system
    mol4 = systems.water_multichannel(lmax=4)  # s .. g channels: the Legendre table's last entry (eval_ecp.py:203-225)
    wf4 = helpers.oracle_wf(mol4, systems.random_mf(mol4))
    wf4.recompute(configs)
    for naip in (None, 26, 50):
        tag = f"l4_naip{naip}"
        en = oenergy.energy(mol4, configs, wf4, 10.0, g[tag + "_rot"], g[tag + "_unif"], naip=naip)
        assert relerr(en["ecp"], g[tag + "_ecp"]) < 1e-10, tag
    assert relerr(g["l4_naip50_ecp"], g["naip50_thr10_ecp"]) > 1e-6  # the f and g channels contribute


def test_g34_batched_ecp():
    """jax_ecp.ECPAccumulator (jax_ecp.py:72-135) and EnergyAccumulator(use_old_ecp=False): energies and T-move tables for
    the default selection, a cut inside one atom's points, no down-selection, and the unrotated grid."""
    from oracle import ecp_batched as ob

    g = golden("g34_ecp_batched")
    mol = systems.water_multichannel()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g["configs"].copy())
    wf.recompute(configs)
    assert np.array_equal(ob.default_naip(mol), g["default_naip"])
    for tag in ("default", "sel6_3", "all", "fixedgrid"):
        naip, nsd, nsr = g[tag + "_naip"], int(g[tag + "_nsd"]), int(g[tag + "_nsr"])
        rot = g[tag + "_rot"] if tag != "fixedgrid" else np.broadcast_to(np.eye(3), g[tag + "_rot"].shape)
        val = ob.ecp(mol, configs, wf, naip, nsd, nsr, rot, g[tag + "_unif"])
        assert relerr(val, g[tag + "_ecp"]) < 1e-10, tag
        for e in (1, 5):
            r = g[f"{tag}_tm{e}_rot"] if tag != "fixedgrid" else np.broadcast_to(np.eye(3), (3, 3, 3))
            d = ob.tmoves(mol, configs, wf, e, 0.02, naip, nsd, nsr, r, g[f"{tag}_tm{e}_unif"])
            for k in ("ratio", "weight", "epos"):
                assert relerr(d[k], g[f"{tag}_tm{e}_{k}"]) < 1e-10, (tag, e, k)
    # where the cut does not split an atom's points the unstable sort the reference calls gives the same energies
    assert relerr(g["default_ecp_default_sort"], g["default_ecp"]) < 1e-13
    val = ob.ecp(mol, configs, wf, g["default_naip"], 12, 1, g["energy_rot"], g["energy_unif"])
    assert relerr(val, g["energy_ecp"]) < 1e-10
    en = oenergy.energy(mol, configs, wf, 10.0, np.zeros((8, 3, 3, 3)), np.ones((8, 3, len(val))))  # (ECP part masked out)
    assert relerr(en["ke"] + en["ee"] + en["ei"] + val + oenergy.coulomb(mol, configs)[2], g["energy_total"]) < 1e-9


def test_g35_more_than_64_electrons_per_spin():
    """(H2O)18, 72 + 72 electrons (slater.py:155-260 takes any number): the oracle on the reference's protocol triangle and on its
    recorded VMC sweep."""
    mol, mf, _, g = helpers.case("g35_big")
    wf = helpers.oracle_wf(mol, mf)
    err = helpers.run_protocol(wf, g, relerr=helpers.relerr_elem)
    bad = {k: v for k, v in err.items() if not v < helpers.g5_tolerance(k, 1e2)}
    assert not bad, bad
    wf = helpers.oracle_wf(mol, mf)
    rec = []
    blk, configs = ovmc.vmc_worker(mol, wf, OpenConfigs(g["vmc_start"].copy()), float(g["vmc_tstep"]), g["vmc_gauss"], g["vmc_unif"],
                                   g["vmc_ecp_rot"], g["vmc_ecp_unif"], record=rec)
    assert np.array_equal(np.asarray(rec).reshape(g["vmc_accepts"].shape), g["vmc_accepts"])
    assert np.max(np.abs(configs.configs - g["vmc_final"])) < 1e-10  # (drift through 72 x 72 inverses after up to 143 rank-1 updates)
    for k in ("energytotal", "energyke", "energyecp", "acceptance"):
        assert abs(blk[k] - g["vmc_blk_" + k]) < 1e-8 * max(1.0, abs(g["vmc_blk_" + k])), k


def test_g9_ecp_ea_detail():
    g = golden("g10_energy")
    mol = systems.water()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g["h2o_configs"].copy())
    wf.recompute(configs)
    d = oenergy.ecp_ea(mol, configs, wf, 1, 0, 10.0, g["ea_rot"], g["ea_unif"])
    assert np.array_equal(d["mask"], g["ea_mask"]) and d["mask"].any() and not d["mask"].all()
    for k in ("total", "local", "ratio", "v_l", "P_l", "epos"):
        assert relerr(d[k], g["ea_" + k]) < 1e-10, k


@pytest.mark.parametrize("tag,mol", [("h2o", systems.water()), ("he", systems.helium())])
def test_g11_vmc_trajectory(tag, mol):
    g = golden("g11_vmc")
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g[tag + "_start"].copy())
    rec = []
    blk, configs = ovmc.vmc_worker(mol, wf, configs, float(g[tag + "_tstep"]), g[tag + "_gauss"], g[tag + "_unif"],
                                   g[tag + "_ecp_rot"], g[tag + "_ecp_unif"], record=rec)
    acc = np.asarray(rec).reshape(g[tag + "_accepts"].shape)
    assert np.array_equal(acc, g[tag + "_accepts"])
    assert 0.05 < acc.mean() < 0.999
    assert relerr(configs.configs, g[tag + "_final"]) < 1e-9
    assert relerr(wf.value()[1], g[tag + "_final_log"]) < 1e-9
    for k in ("energyke", "energyee", "energyei", "energyecp", "energygrad2", "energytotal", "acceptance"):
        assert relerr(blk[k], g[f"{tag}_blk_{k}"]) < 1e-8, k
    # output-dict contract of the reference's vmc_worker (C1 plumbing)
    assert set(g[tag + "_blk_keys"].tolist()) == set(blk.keys())


def test_g37_vmc_trajectory_of_the_headline_system():
    """(H2O)8 — 64 electrons, the system of BASELINE.json's metric: one vmc_worker sweep of 4 walkers with the energy, generated by the
    reference (make_golden.g_vmc_cluster); the oracle replays it on the recorded draws."""
    g = golden("g37_vmc_cluster")
    mol = systems.water_cluster()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    configs = OpenConfigs(g["start"].copy())
    rec = []
    blk, configs = ovmc.vmc_worker(mol, wf, configs, float(g["tstep"]), g["gauss"], g["unif"], g["ecp_rot"], g["ecp_unif"], record=rec)
    acc = np.asarray(rec).reshape(g["accepts"].shape)
    assert np.array_equal(acc, g["accepts"]) and 0.05 < acc.mean() < 0.999
    assert relerr(configs.configs, g["final"]) < 1e-9
    assert np.max(np.abs(wf.value()[1] - g["final_log"])) < 1e-8
    for k in ("energyke", "energyee", "energyei", "energyecp", "energygrad2", "energytotal", "acceptance"):
        assert relerr(blk[k], g[f"blk_{k}"]) < 1e-8, k


def test_g12_dmc_propagate_and_branch():
    """dmc_propagate (dmc.py:123-221: T-moves, drift-diffusion with fixed-node rejection, weights) and branch
    (:342-376) replayed with the reference's own random draws."""
    from oracle import dmc as odmc

    g = golden("g12_dmc")
    mol = systems.water()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    rec = []
    df, configs, weights = odmc.dmc_propagate(mol, wf, OpenConfigs(g["start"].copy()), g["weights0"].copy(), float(tstep),
                                              float(branchcut), float(e_trial), float(e_est), int(nsteps),
                                              helpers.ReplayTape(g), record=rec)
    acc = np.asarray([r[2] for r in rec])
    assert np.array_equal(acc, g["accepts"]) and acc[:8].any()  # some T-moves were accepted
    assert relerr(configs.configs, g["final"]) < 1e-10 and relerr(weights, g["weights"]) < 1e-9
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert relerr(df[k], g["df_" + k]) < 1e-9, k
    newinds, wnew, info = odmc.branch(g["branch_configs"], g["branch_weights"], float(g["branch_u"]))
    assert np.array_equal(g["branch_configs"][newinds], g["branch_newconfigs"])
    assert relerr(wnew, g["branch_newweights"]) < 1e-14
    assert [info["max branches"], info["Number of walkers killed"]] == g["branch_info"].tolist()


def test_g7_three_body_jastrow_multidet():
    """ThreeBodyJastrow (three_body_jastrow.py:19-655) alone and inside the 12-determinant x 2-body x 3-body
    product (BASELINE config C4 shape): protocol triangle, EnergyAccumulator dict, vmc_worker trajectory."""
    import ast

    g = golden("g7_jastrow3_multidet")
    mol = systems.water()
    mf = systems.random_mf(mol, nvirt=6)
    dets = ast.literal_eval(str(g["det_json"]))
    wf = helpers.oracle_wf3(mol, mf, dets)
    err = helpers.run_protocol3(wf, g)
    bad = {k: v for k, v in err.items() if v > 1e-10}
    assert not bad, bad
    configs = OpenConfigs(g["final_configs"].copy())
    wf.recompute(configs)
    en = oenergy.energy(mol, configs, wf, 10.0, g["energy_rot"], g["energy_unif"])
    for k in ("ke", "ee", "ei", "ecp", "grad2", "total"):
        assert relerr(en[k], g["energy_" + k]) < 1e-9, k
    rec = []
    blk, cfg = ovmc.vmc_worker(mol, wf, OpenConfigs(g["vmc_start"].copy()), 0.3, g["vmc_gauss"], g["vmc_unif"], g["vmc_ecp_rot"],
                               g["vmc_ecp_unif"], record=rec)
    assert np.array_equal(np.asarray(rec).reshape(g["vmc_accepts"].shape), g["vmc_accepts"])
    assert relerr(cfg.configs, g["vmc_final"]) < 1e-9
    for k in ("energyke", "energyecp", "energytotal", "acceptance"):
        assert relerr(blk[k], g["vmc_blk_" + k]) < 1e-8, k


def test_testvalue_many_matches_reference():
    """testvalue_many (used by the density-matrix accumulators) on the 12-determinant Slater x 2-body x 3-body H2O wave
    function and on a periodic diamond supercell (tests/golden/g18_testvalue_many.npz)."""
    import ast

    from helpers import oracle_pbc_wf, oracle_wf3
    from pyqmc_amd.configs import OpenConfigs, PeriodicConfigs

    g = golden("g18_testvalue_many")
    g7 = golden("g7_jastrow3_multidet")
    mol = systems.water()
    wf = oracle_wf3(mol, systems.random_mf(mol, nvirt=6), ast.literal_eval(str(g7["det_json"])))
    cfg = OpenConfigs(g["h2o_configs"].copy())
    wf.recompute(cfg)
    epos = cfg.make_irreducible(0, g["h2o_aux"])
    for nm, w in (("slater", wf.wf_factors[0]), ("j2", wf.wf_factors[1]), ("j3", wf.wf_factors[2]), ("wf", wf)):
        assert relerr(w.testvalue_many(g["h2o_es"], epos), g[f"h2o_{nm}"]) < 1e-10, nm
    pg = wf.pgradient()  # parameter gradients of the same state (slater.py:462-542, jastrowspin.py:457-464, three_body_jastrow.py:657-719)
    assert sorted(pg.keys()) == g["h2o_pgrad_keys"].tolist()
    for k, v in pg.items():
        assert relerr(v, g["h2o_pgrad_" + k]) < 1e-10, k
    sup, pwf = oracle_pbc_wf("fcc2cubic")
    cfg = PeriodicConfigs(g["pbc_configs"].copy(), sup.lattice_vectors(), wrap=g["pbc_wrap"].copy())
    pwf.recompute(cfg)
    epos = cfg.make_irreducible(0, g["pbc_aux"])
    for nm, w in (("slater", pwf.wf_factors[0]), ("j2", pwf.wf_factors[1]), ("wf", pwf)):
        assert relerr(w.testvalue_many(g["pbc_es"], epos), g[f"pbc_{nm}"]) < 1e-9, nm


def test_compiled_ao_backend_of_the_oracle_matches_the_numpy_routine():
    """oracle/ao_eval.c (the CPU baseline's AO evaluator: the same loops compiled) against oracle/gto.py's NumPy
    routine — itself pinned to the reference by g2 — on the metric system and on H2O: value, gradient, Laplacian."""
    from oracle import gto

    rng = np.random.default_rng(3)
    try:
        for mol in (systems.water_cluster(), systems.water()):
            tab = gto.AOTable(mol)
            pts = mol.atom_coords()[rng.integers(mol.natm, size=300)] + rng.standard_normal((300, 3)) * 1.5
            for ncomp in (1, 4, 5):
                gto.set_ao_backend("numpy")
                a = gto.eval_ao(tab, pts, ncomp)
                gto.set_ao_backend("c")
                b = gto.eval_ao(tab, pts, ncomp)
                assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-13 * max(1.0, np.max(np.abs(a)))
    finally:
        gto.set_ao_backend("numpy")
