"""GPU parity of the periodic-boundary pieces of the hot path against outputs of the real reference
(tests/golden/g14_pbc_jastrow.npz; generator tests/golden/make_golden.py:g_pbc).  All calls go through the C ABI."""

import numpy as np
import pytest

import helpers
from helpers import PBC_JASTROW_CASES, golden, pbc_jastrow_coeffs, run_protocol_pbc
from pyqmc_amd import pbc, systems

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["cubic", "prim"])
def test_periodic_jastrow_matches_reference(tag):
    """JastrowSpin on PeriodicConfigs: minimal-image e-e / e-ion displacements inside the kernels
    (diagonal cell: folded fractional coordinates; fcc primitive cell: 27-image rule, cut-off beyond half the
    plane spacing so that the rule matters)."""
    import pyqmc_amd as pa

    g = golden("g14_pbc_jastrow")
    make, kws = PBC_JASTROW_CASES[tag]
    cell = make()
    ja, _ = pa.wf.generate_jastrow(cell, **kws)
    a, b = pbc_jastrow_coeffs(cell)
    ja.parameters["acoeff"], ja.parameters["bcoeff"] = a, b
    err = run_protocol_pbc({"jastrow": ja}, g, f"{tag}_", cell)
    assert max(err.values()) < 1e-10, {k: v for k, v in err.items() if v > 1e-11}


def test_periodic_three_body_jastrow_matches_reference():
    import pyqmc_amd as pa

    g = golden("g14_pbc_jastrow")
    cell = systems.diamond_primitive()
    j3, _ = pa.wf.generate_jastrow3(cell, rcut=3.0)
    j3.parameters["ccoeff"] = g["prim3_ccoeff"]
    err = run_protocol_pbc({"j3": j3}, g, "prim3_", cell, update_first=True)
    assert max(err.values()) < 1e-9, {k: v for k, v in err.items() if v > 1e-10}


# ------------------------------------------------------------------ periodic orbitals and Slater determinants
@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic"])
def test_periodic_orbitals_match_reference(tag):
    """Lattice-summed AOs and Bloch MOs (numba/pbcgto.py + PBCOrbitalEvaluatorKpoints) against the reference's
    numbers: the device evaluates Gamma-point AOs of the supercell; unfolding them with the copies' Bloch phases must
    give the reference's per-k AOs (wrap phase included), and the folded coefficients its MOs."""
    import pyqmc_amd as pa
    from helpers import pbc_slater_case, unfold_ao

    g = golden("g15_pbc_orbitals")
    sup, mf = pbc_slater_case(tag)
    sl = pa.Slater(sup, mf)
    pts = g[f"{tag}_pts"].reshape(-1, 3)
    for nm, nc in (("val", 1), ("grad", 4), ("lap", 5)):
        ao = sl._dev.eval_ao(pts, nc)  # (nc, P, nao_super)
        ref = g[f"{tag}_ao_{nm}"]
        ref = ref.reshape((ref.shape[0], nc, -1, ref.shape[-1]))
        assert helpers.relerr(unfold_ao(sup, mf.kpts, ao), ref) < 1e-12, nm
    for nm, nc in (("val", 1), ("lap", 5)):
        ref = g[f"{tag}_mo_{nm}"]
        ref = ref.reshape((nc, -1, ref.shape[-1]))
        for mfma in (True, False):
            mo = sl._dev.eval_mo(0, pts, nc, use_mfma=mfma)
            assert helpers.relerr(mo, ref) < 1e-12, (nm, mfma)
    # the translation-invariant rule (every image inside the cut-offs) differs from the reference's truncated list
    # only at the level of the terms the reference drops (make_golden prints ~5e-4 for the Laplacian)
    full = pa.Slater(sup, mf, image_rule="complete")
    ao = unfold_ao(sup, mf.kpts, full._dev.eval_ao(pts, 5))
    ref = g[f"{tag}_ao_lap_allLs"]
    d_all = helpers.relerr(ao, ref.reshape(ao.shape))
    assert d_all < 1e-12, d_all


@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic", "k222"])
def test_periodic_slater_jastrow_matches_reference(tag):
    """Slater x Jastrow on PeriodicConfigs (diamond; Gamma cell, 4-fold non-diagonal supercell, 2x2x2 supercell =
    64 electrons with 8 k-points) through the whole protocol against the reference."""
    import pyqmc_amd as pa
    from helpers import pbc_slater_case

    g = golden("g15_pbc_orbitals")
    sup, mf = pbc_slater_case(tag)
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    err = run_protocol_pbc({"slater": wf.wf_factors[0], "jastrow": wf.wf_factors[1], "wf": wf}, g, f"{tag}_", sup)
    assert max(err.values()) < 2e-9, {k: v for k, v in err.items() if v > 1e-10}


# ------------------------------------------------------------------ periodic energies and VMC
@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic"])
def test_periodic_energy_matches_reference(tag):
    """EnergyAccumulator on a periodic cell: Ewald ee / ei / ii (observables/ewald.py), kinetic energy through the
    lattice-summed orbitals and the ECP integrator with minimal-image electron-ion vectors and folded quadrature
    points, against the reference with its own random draws replayed."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g16_pbc_energy")
    sup, wf = helpers.gpu_pbc_wf(tag)
    cfg = PeriodicConfigs(g[f"{tag}_configs"].copy(), sup.lattice_vectors())
    wf.recompute(cfg)
    for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
        en = pa.EnergyAccumulator(sup, threshold=thr, ewald_gmax=10)(cfg, wf, rot=g[f"{tag}_{thr_tag}_rot"], unif=g[f"{tag}_{thr_tag}_unif"])
        for k in ("ke", "ee", "ei", "ecp", "grad2", "total"):
            assert helpers.relerr(en[k], g[f"{tag}_{thr_tag}_{k}"]) < 2e-9, (thr_tag, k)
    # Ewald alone, default reciprocal range
    en = pa.EnergyAccumulator(sup, threshold=-1.0)(cfg, wf, rot=g[f"{tag}_det_rot"], unif=g[f"{tag}_det_unif"])
    assert helpers.relerr(en["ee"], g[f"{tag}_ewald_ee"]) < 1e-12 and helpers.relerr(en["ei"], g[f"{tag}_ewald_ei"]) < 1e-12


def test_periodic_cell_with_128_electrons_per_spin_matches_oracle():
    """64-atom diamond cell at Gamma: 128 + 128 electrons, 832 AOs (slater.py:155-260, orbitals.py:192-239 take any size; here the
    general-n kernels: lattice-summed AOs by the thread-per-point evaluator, k_mo_rows, two columns per lane in the determinant
    kernels).  Recompute, derivatives of the first / a middle / the last electron, an accepted update and the local energy of 2
    walkers against the oracle."""
    import pyqmc_amd as pa
    from oracle import energy as oenergy
    from pyqmc_amd.configs import PeriodicConfigs

    sup, wf = helpers.gpu_pbc_wf("big")
    _, owf = helpers.oracle_pbc_wf("big")
    assert sup.nelec == (128, 128)
    x = pa.initial_guess(sup, 2, rng=np.random.default_rng(128)).configs
    cfg, ocfg = PeriodicConfigs(x.copy(), sup.lattice_vectors()), PeriodicConfigs(x.copy(), sup.lattice_vectors())
    (s1, l1), (s2, l2) = wf.recompute(cfg), owf.recompute(ocfg)
    assert np.array_equal(s1, s2) and note("big_pbc:recompute_log", float(np.max(np.abs(l1 - l2)))) < 1e-8
    rng = np.random.default_rng(3)
    for e in (0, 127, 128, 255):
        new = x[:, e, :] + 0.2 * rng.standard_normal((2, 3))
        ep, oep = cfg.make_irreducible(e, new), ocfg.make_irreducible(e, new)
        (g1, v1, sv), (g2, v2, osv) = wf.gradient_value(e, ep), owf.gradient_value(e, oep)
        assert note(f"big_pbc:e{e}_val", helpers.relerr(v1, v2)) < 1e-8 and note(f"big_pbc:e{e}_grad", helpers.relerr(g1, g2)) < 1e-7
        (g1, p1), (g2, p2) = wf.gradient_laplacian(e, ep), owf.gradient_laplacian(e, oep)
        assert note(f"big_pbc:e{e}_lap", helpers.relerr(p1, p2)) < 1e-7
        acc = np.array([True, e % 2 == 0])
        cfg.move(e, ep, acc)
        ocfg.move(e, oep, acc)
        wf.updateinternals(e, ep, cfg, mask=acc, saved_values=sv)
        owf.updateinternals(e, oep, ocfg, mask=acc, saved_values=osv)
    assert note("big_pbc:value_after_updates", float(np.max(np.abs(wf.value()[1] - owf.value()[1])))) < 1e-8
    N, necp = 256, 64
    rot = pa.ecp_batched.random_rotations(N * necp).reshape(N, necp, 3, 3)
    unif = rng.random((N, necp, 2))
    en = pa.EnergyAccumulator(sup, ewald_gmax=6)(cfg, wf, rot=rot, unif=unif)
    oen = oenergy.energy(sup, ocfg, owf, 10.0, rot, unif, ewald_kws=dict(ewald_gmax=6))
    for k in ("ke", "ee", "ei", "ecp", "total"):
        assert note(f"big_pbc:energy_{k}", helpers.relerr(en[k], oen[k])) < 1e-7, k


def test_periodic_cell_with_108_complex_orbitals_per_spin_matches_oracle():
    """Diamond, 3 x 3 x 3 primitive cells (54 atoms, 108 + 108 electrons, complex Bloch orbitals at 27 k-points): complex determinants
    beyond one column per lane — k_build_invert_cg (elimination on a scratch matrix in global memory: the complex tile is 188 KB),
    slater_ratios_c, Sherman-Morrison on the inverse in place — with the thread-per-point lattice sums + k_mo_rows.  Recompute,
    derivatives of the first / a middle / the last electron, accepted updates, kinetic + ECP energy of 2 walkers against the oracle."""
    import pyqmc_amd as pa
    from oracle import energy as oenergy
    from pyqmc_amd.configs import PeriodicConfigs

    sup, wf = helpers.gpu_pbc_wf("big_complex")
    _, owf = helpers.oracle_pbc_wf("big_complex")
    assert sup.nelec == (108, 108) and wf.dtype == complex
    x = pa.initial_guess(sup, 2, rng=np.random.default_rng(108)).configs
    cfg, ocfg = PeriodicConfigs(x.copy(), sup.lattice_vectors()), PeriodicConfigs(x.copy(), sup.lattice_vectors())
    (s1, l1), (s2, l2) = wf.recompute(cfg), owf.recompute(ocfg)
    assert note("big_complex:recompute_phase", float(np.max(np.abs(s1 - s2)))) < 1e-8 and note("big_complex:recompute_log", float(np.max(np.abs(l1 - l2)))) < 1e-8
    rng = np.random.default_rng(4)
    for e in (0, 107, 108, 215):
        new = x[:, e, :] + 0.2 * rng.standard_normal((2, 3))
        ep, oep = cfg.make_irreducible(e, new), ocfg.make_irreducible(e, new)
        (g1, v1, sv), (g2, v2, osv) = wf.gradient_value(e, ep), owf.gradient_value(e, oep)
        assert note(f"big_complex:e{e}_val", helpers.relerr(v1, v2)) < 1e-8 and note(f"big_complex:e{e}_grad", helpers.relerr(g1, g2)) < 1e-7
        (g1, p1), (g2, p2) = wf.gradient_laplacian(e, ep), owf.gradient_laplacian(e, oep)
        assert note(f"big_complex:e{e}_lap", helpers.relerr(p1, p2)) < 1e-7
        acc = np.array([True, e % 2 == 0])
        cfg.move(e, ep, acc)
        ocfg.move(e, oep, acc)
        wf.updateinternals(e, ep, cfg, mask=acc, saved_values=sv)
        owf.updateinternals(e, oep, ocfg, mask=acc, saved_values=osv)
    (s1, l1), (s2, l2) = wf.value(), owf.value()
    assert note("big_complex:value_after_updates", float(max(np.max(np.abs(l1 - l2)), np.max(np.abs(s1 - s2))))) < 1e-8
    N, necp = 216, 54
    rot = pa.ecp_batched.random_rotations(N * necp).reshape(N, necp, 3, 3)
    unif = rng.random((N, necp, 2))
    en = pa.EnergyAccumulator(sup, ewald_gmax=6)(cfg, wf, rot=rot, unif=unif)
    oen = oenergy.energy(sup, ocfg, owf, 10.0, rot, unif, ewald_kws=dict(ewald_gmax=6))
    for k in ("ke", "ee", "ei", "ecp", "total"):
        assert note(f"big_complex:energy_{k}", helpers.relerr(en[k], oen[k])) < 1e-7, k
    # one fused VMC sweep leaves a state that equals a fresh recompute
    dev = wf.fused_device()
    dev.vmc_sweeps(0.3, 1, seed=3, energy=False)
    xs = dev.configs()
    l_upd = dev.value()[1]
    assert note("big_complex:sweep_update_vs_recompute", float(np.max(np.abs(dev.recompute(xs)[1] - l_upd)))) < 1e-8


@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic"])
def test_periodic_batched_ecp_matches_oracle(tag):
    """The batched ECP integrator (jax_ecp.py:72-135; oracle/ecp_batched.py is pinned to the reference by g34) in a periodic cell:
    minimal-image electron-ion vectors, every carbon's 6 points in one table per electron, 6 kept + 1 sampled; energies and the
    T-move table of one electron against the oracle with the same rotations and selection uniforms."""
    import pyqmc_amd as pa
    from oracle import ecp_batched as ob
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g16_pbc_energy")
    sup, wf = helpers.gpu_pbc_wf(tag)
    _, owf = helpers.oracle_pbc_wf(tag)
    x = g[f"{tag}_configs"].copy()
    cfg, ocfg = PeriodicConfigs(x.copy(), sup.lattice_vectors()), PeriodicConfigs(x.copy(), sup.lattice_vectors())
    wf.recompute(cfg)
    owf.recompute(ocfg)
    W, N = x.shape[:2]
    acc = pa.ECPAccumulator(sup)
    necp = len(acc.naip)
    assert acc.nselect_deterministic == 6 and acc.nselect_random == 1 and acc.naip.sum() == 6 * necp
    rng = np.random.default_rng(16)
    rot = pa.ecp_batched.random_rotations(N * necp).reshape(N, necp, 3, 3)
    unif = rng.random((N, W, 1))
    val = acc(cfg, wf, rot=rot, unif=unif)
    ref = ob.ecp(sup, ocfg, owf, acc.naip, 6, 1, rot, unif)
    assert helpers.relerr(val, ref) < 2e-9
    d = acc.nonlocal_tmoves(cfg, wf, 2, 0.01, rot=rot[2], unif=unif[2])
    o = ob.tmoves(sup, ocfg, owf, 2, 0.01, acc.naip, 6, 1, rot[2], unif[2])
    assert helpers.relerr(d["weight"], o["weight"]) < 1e-9 and helpers.relerr(d["ratio"], o["ratio"]) < 2e-9
    wrapped = PeriodicConfigs(x.copy(), sup.lattice_vectors()).make_irreducible(2, o["epos"])
    assert np.max(np.abs(d["configs"].configs - wrapped.configs)) < 1e-10


@pytest.mark.parametrize("fused", [True, False])
def test_periodic_vmc_trajectory_matches_reference(fused, monkeypatch):
    """vmc_worker on PeriodicConfigs replayed with the reference's random draws: identical accept decisions, final
    folded coordinates AND wrap counters, block energies — fused device sweep and protocol path."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g16_pbc_energy")
    sup, wf = helpers.gpu_pbc_wf("gamma")
    cfg = PeriodicConfigs(g["vmc_start"].copy(), sup.lattice_vectors(), wrap=g["vmc_start_wrap"].copy())
    tstep, nsteps = float(g["vmc_tstep"]), int(g["vmc_nsteps"])
    acc = pa.EnergyAccumulator(sup, ewald_gmax=10)
    if fused:
        tapes = dict(gauss=g["vmc_gauss"], unif=g["vmc_unif"], ecp_rot=g["vmc_ecp_rot"], ecp_unif=g["vmc_ecp_unif"], record=[])
        blk, cfg = pa.vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc}, tapes=tapes)
        accepts = tapes["record"][0]
    else:
        gz, un = iter(g["vmc_gauss"].reshape(-1, *g["vmc_gauss"].shape[2:])), iter(g["vmc_unif"].reshape(-1, g["vmc_unif"].shape[-1]))
        monkeypatch.setattr(np.random, "normal", lambda scale, size: scale * next(gz))
        monkeypatch.setattr(np.random, "rand", lambda n: next(un))
        rots, eun = iter(g["vmc_ecp_rot"]), iter(g["vmc_ecp_unif"])
        accepts = []
        orig = wf.updateinternals
        monkeypatch.setattr(wf, "updateinternals", lambda e, ep, c, mask=None, saved_values=None: (accepts.append(mask.copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1])
        monkeypatch.setattr(acc, "avg", lambda c, w: {k: np.mean(v) for k, v in acc(c, w, rot=next(rots), unif=next(eun)).items()})
        blk, cfg = helpers.protocol_vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc})
        accepts = np.asarray(accepts).reshape(g["vmc_accepts"].shape)
    assert np.array_equal(np.asarray(accepts, dtype=bool), g["vmc_accepts"])
    assert helpers.relerr(cfg.configs, g["vmc_final"]) < 1e-9 and np.array_equal(cfg.wrap, g["vmc_final_wrap"])
    assert helpers.relerr(wf.value()[1], g["vmc_final_log"]) < 1e-9
    for k in ("energyke", "energyee", "energyei", "energyecp", "energygrad2", "energytotal", "acceptance"):
        assert helpers.relerr(blk[k], g[f"vmc_blk_{k}"]) < 1e-8, k


def test_periodic_fused_sweep_lane_and_wave_paths_agree(monkeypatch):
    """64-electron 2x2x2 diamond supercell (config C5 shape): the lane-per-walker and the wave-per-walker fused sweeps
    take the same decisions, keep the walkers inside the cell and agree with a fresh recompute."""
    import pyqmc_amd as pa

    sup, mf = helpers.pbc_slater_case("k222")
    start = pa.initial_guess(sup, 200, rng=np.random.default_rng(5))
    res = []
    for lw in ("1", "0"):
        monkeypatch.setenv("PQA_LW", lw)
        _, wf = helpers.gpu_pbc_wf("k222")
        dev = wf.fused_device()
        cfg = start.copy()
        wf.recompute(cfg)
        acc, en, rec = dev.vmc_sweeps(0.3, 1, seed=77, energy=True, record=True)
        x = dev.configs()
        frac = x @ np.linalg.inv(sup.lattice_vectors())
        assert frac.min() >= -1e-12 and frac.max() < 1 + 1e-12
        res.append((x, dev.value()[1], en, rec, dev.recompute(x)[1], dev.wrap_delta()))
    same = res[0][3] == res[1][3]
    assert same.mean() > 0.9999
    ok = same.all(axis=(0, 1))
    assert helpers.relerr(res[0][0][ok], res[1][0][ok]) < 1e-10 and np.array_equal(res[0][5][ok], res[1][5][ok])
    assert np.abs(res[0][5]).sum() > 0  # some walkers did cross the boundary
    for r in res:
        assert np.max(np.abs(r[1] - r[4])) < 1e-8


def test_periodic_dmc_propagate_matches_reference():
    """dmc_propagate (dmc.py:123-221) on a periodic cell with every wave-function / Ewald / ECP / T-move quantity from
    the HIP library, replaying the reference's random draws: identical decisions, coordinates, wrap counters, weights."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g17_pbc_dmc")
    sup, wf = helpers.gpu_pbc_wf("gamma")
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    accepts = []
    orig = wf.updateinternals
    wf.updateinternals = lambda e, ep, c, mask=None, saved_values=None: (accepts.append(np.asarray(mask).copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = helpers.protocol_dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                        nsteps=int(nsteps), accumulators={"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)},
                                        rng=helpers.ReplayTape(g))
    assert np.array_equal(np.asarray(accepts), g["accepts"])
    assert helpers.relerr(cfg.configs, g["final"]) < 1e-9 and np.array_equal(cfg.wrap, g["final_wrap"])
    assert helpers.relerr(weights, g["weights"]) < 1e-8
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert helpers.relerr(df[k], g["df_" + k]) < 1e-8, k


def test_periodic_fused_dmc_steps_match_reference():
    """pqa_dmc_steps on the periodic cell (T-move candidates folded into the cell, wrap counters, Ewald energies)."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g17_pbc_dmc")
    sup, wf = helpers.gpu_pbc_wf("gamma")
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = pa.dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                        nsteps=int(nsteps), accumulators={"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)},
                                        rng=helpers.ReplayTape(g))
    assert helpers.relerr(cfg.configs, g["final"]) < 1e-9 and np.array_equal(cfg.wrap, g["final_wrap"])
    assert helpers.relerr(weights, g["weights"]) < 1e-8
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert helpers.relerr(df[k], g["df_" + k]) < 1e-8, k


def note(key, val):  # recorded with the other measured errors (gpurun_out/parity_report.json, written by test_gpu_parity's fixture)
    from test_gpu_parity import REPORT

    REPORT["pbc_" + key] = float(val)
    return float(val)


def _complex_dmc_wf():
    import pyqmc_amd as pa
    from helpers import pbc_complex_case

    sup, mf = pbc_complex_case()
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    assert wf.dtype == complex
    return sup, wf


def test_complex_dmc_propagate_protocol_route_matches_reference():
    """Complex wave function (3x1x1 diamond supercell, complex Bloch coefficients), dmc_propagate with ECP T-moves through the
    PROTOCOL entry points of the device wave function (tests/helpers.protocol_dmc_propagate: the reference's control flow;
    EnergyAccumulator.nonlocal_tmoves takes candidate positions / weights from pqa_tmoves and the complex ratios from
    testvalue): the reference's run of golden g30 — decisions of the T-move and drift-diffusion phases, walkers, wrap counters,
    weights, complex block averages.  T-move amplitudes from Re[Psi(R')/Psi(R)] (make_golden.py:g_pbc_complex_dmc)."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g30_pbc_complex_dmc")
    sup, wf = _complex_dmc_wf()
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    accepts = []
    orig = wf.updateinternals
    wf.updateinternals = lambda e, ep, c, mask=None, saved_values=None: (accepts.append(np.asarray(mask).copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = helpers.protocol_dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                                      nsteps=int(nsteps), accumulators={"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)},
                                                      rng=helpers.ReplayTape(g))
    assert np.array_equal(np.asarray(accepts), g["accepts"])
    assert note("cdmc_final", helpers.relerr(cfg.configs, g["final"])) < 1e-9 and np.array_equal(cfg.wrap, g["final_wrap"])
    assert note("cdmc_weights", helpers.relerr(weights, g["weights"])) < 1e-8
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert helpers.relerr(df[k], g["df_" + k]) < 1e-8, k


def test_complex_fused_dmc_steps_match_reference():
    """pqa_dmc_steps with complex determinants (round 3): the whole step loop on the device — complex T-move ratios and
    Sherman-Morrison commits in k_tm_walker<CX>, complex lane-per-walker drift-diffusion without a node constraint, complex ECP
    energies, weights from Re E_L — replaying the reference's draws of golden g30: walkers, wrap counters, weights and the
    (complex) block averages."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g30_pbc_complex_dmc")
    sup, wf = _complex_dmc_wf()
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = pa.dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                        nsteps=int(nsteps), accumulators={"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)},
                                        rng=helpers.ReplayTape(g))
    assert note("cfdmc_final", helpers.relerr(cfg.configs, g["final"])) < 1e-9 and np.array_equal(cfg.wrap, g["final_wrap"])
    assert note("cfdmc_weights", helpers.relerr(weights, g["weights"])) < 1e-8
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert note("cfdmc_" + k, helpers.relerr(df[k], g["df_" + k])) < 1e-8, k
    assert abs(np.imag(df["energytotal"])) > 1e-3


def test_complex_fused_dmc_on_twisted_cell_is_consistent():
    """Twisted cell (the handle keeps true, unfolded coordinates): the fused complex DMC step is reproducible from its seed,
    leaves the walkers inside the cell with consistent wrap counters, and its updated state equals a fresh recompute."""
    import pyqmc_amd as pa
    from helpers import twist_case

    sup, mf = twist_case("s211")
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    acc = {"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)}
    W = 64
    start = pa.initial_guess(sup, W, rng=np.random.default_rng(5))
    outs = []
    for rep in range(2):
        np.random.seed(77)
        cfg = start.copy()
        df, cfg, w = pa.dmc_propagate(wf, cfg, np.ones(W), 0.02, 50.0, -20.0, -20.0, nsteps=3, accumulators=acc)
        outs.append((cfg.configs.copy(), cfg.wrap.copy(), w.copy(), df["energytotal"], wf.value()[1].copy()))
    assert all(np.array_equal(p, q) for p, q in zip(outs[0], outs[1]))
    x, wrap, w, e, logv = outs[0]
    frac = x @ np.linalg.inv(sup.lattice_vectors())
    assert frac.min() >= -1e-12 and frac.max() < 1 + 1e-12 and np.all(np.isfinite(w)) and np.iscomplexobj(e)
    fresh = wf.recompute(cfg)[1]
    assert note("cfdmc_twist_update_vs_recompute", np.max(np.abs(fresh - logv))) < 1e-9


def test_periodic_rundmc_smoke():
    import pyqmc_amd as pa

    np.random.seed(11)
    sup, wf = helpers.gpu_pbc_wf("fcc2cubic")
    configs = pa.initial_guess(sup, 128, rng=np.random.default_rng(3))
    df, configs, weights = pa.rundmc(wf, configs, tstep=0.02, nblocks=2, nsteps_per_block=2, vmc_warmup=2,
                                     accumulators={"energy": pa.EnergyAccumulator(sup)})
    assert df["energytotal"].shape == (2,) and np.all(np.isfinite(df["energytotal"]))
    frac = configs.configs @ np.linalg.inv(sup.lattice_vectors())
    assert frac.min() >= -1e-12 and frac.max() < 1 + 1e-12 and configs.wrap.shape == configs.configs.shape


def test_ewald_madelung_constants_on_device():
    """k_ewald on the reference's known-answer systems (NaCl, CaF2 Madelung constants, tests/unit/test_ewald.py) through a
    Jastrow-only periodic handle."""
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    for name, sup, cfg, want in helpers.madelung_cases():
        ja, _ = pa.wf.generate_jastrow(sup)
        c = PeriodicConfigs(cfg, sup.lattice_vectors())
        ja.recompute(c)
        en = pa.EnergyAccumulator(sup)(c, ja)
        coulomb = en["total"] - en["ke"] - en["ecp"]
        assert abs(coulomb[0] - want) < 1e-4 * max(1, abs(want) / 1.7), (name, coulomb, want)


def test_ewald_edge_cases_coincident_particles_and_cutoff_boundary():
    """k_ewald's 1/r comes from v_rsq_f64 + Newton steps and its erfc from a table cut at alpha^2 r^2 = 40.
    (a) two coincident electrons: the pair term is +inf (as erfc(0)/0 was with the IEEE sequence), never NaN; a pair 1e-9 bohr
    apart gives a large finite energy equal to the oracle's.  (b) an electron pair whose distance puts the nearest image EXACTLY
    at alpha r = sqrt(40) (and hair-widths either side): the energy is continuous across the cut (the dropped term is < 4e-19)
    and equals the oracle's Ewald sum."""
    import pyqmc_amd as pa
    from oracle import pbc as opbc
    from pyqmc_amd.configs import PeriodicConfigs

    name, sup, cfg, want = helpers.madelung_cases()[3]  # CaF2 conventional cell: 8 electrons
    ja, _ = pa.wf.generate_jastrow(sup)
    ew = opbc.Ewald(sup)
    acc = pa.EnergyAccumulator(sup)

    def ee_energy(x):  # the electron-electron Ewald energy (the pair sums under test); (the Jastrow's own 1/r terms make ke NaN at r = 0)
        c = PeriodicConfigs(x, sup.lattice_vectors())
        ja.recompute(c)
        with np.errstate(all="ignore"):
            en = acc(c, ja)
        return en["ee"][0], c

    x = cfg.copy()
    x[0, 1] = x[0, 0]  # (a) coincident
    e, _ = ee_energy(x)
    assert np.isinf(e) and e > 0
    x[0, 1] = x[0, 0] + np.array([1e-9, 0.0, 0.0])
    e, c = ee_energy(x)
    ref = ew.energy(c)[0][0]
    assert np.isfinite(e) and abs(e - ref) < 1e-6 * abs(ref)
    rcut = np.sqrt(40.0) / ew.alpha  # (b)
    vals = []
    for d in (rcut * (1 - 1e-12), rcut, rcut * (1 + 1e-12)):
        x = cfg.copy()
        x[0, 1] = x[0, 0] + np.array([d, 0.0, 0.0])
        e, c = ee_energy(x)
        vals.append(e)
        assert abs(e - ew.energy(c)[0][0]) < 1e-9 * max(1.0, abs(e))
    assert max(vals) - min(vals) < 1e-9


# ------------------------------------------------------------------ complex Bloch orbitals
def test_complex_periodic_slater_matches_reference():
    """k-points off the time-reversal-invariant set (3x1x1 diamond supercell: k = 0, 1/3, 2/3 b1, complex coefficients): complex
    MOs, determinant phases, gradients, Laplacians, ratios, Sherman-Morrison updates against the reference
    (tests/golden/g19_pbc_complex.npz) — the orbital kernel runs as a real GEMM on [Re C | Im C], the determinant kernels
    in complex arithmetic (csrc/pqa_cslater.hpp)."""
    import pyqmc_amd as pa
    from helpers import pbc_complex_case

    g = golden("g19_pbc_complex")
    sup, mf = pbc_complex_case()
    wf = pa.generate_wf(sup, mf)
    sl = wf.wf_factors[0]
    assert sl.dtype == complex
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    pts = g["pts"].reshape(-1, 3)
    for nm, nc in (("val", 1), ("lap", 5)):
        ref = g[f"mo_{nm}"]
        mo = sl._dev.eval_mo(0, pts, nc)
        assert helpers.relerr(mo, ref.reshape((nc, -1, ref.shape[-1]))) < 1e-12, nm
    err = run_protocol_pbc({"slater": sl, "jastrow": wf.wf_factors[1], "wf": wf}, g, "", sup)
    assert max(err.values()) < 2e-9, {k: v for k, v in err.items() if v > 1e-10}
    # complex local energies: ecp and total complex, the rest real (accumulators.py:60-75 with eval_ecp.py:89)
    from pyqmc_amd.configs import PeriodicConfigs

    cfg = PeriodicConfigs(g["en_configs"].copy(), sup.lattice_vectors())
    wf.recompute(cfg)
    for thr_tag, thr in (("det", -1.0), ("thr10", 10.0)):
        en = pa.EnergyAccumulator(sup, threshold=thr, ewald_gmax=10)(cfg, wf, rot=g[f"en_{thr_tag}_rot"], unif=g[f"en_{thr_tag}_unif"])
        for k in ("ke", "ee", "ei", "ecp", "grad2", "total"):
            assert en[k].dtype == g[f"en_{thr_tag}_{k}"].dtype, k
            assert helpers.relerr(en[k], g[f"en_{thr_tag}_{k}"]) < 2e-9, (thr_tag, k)


@pytest.mark.parametrize("fused", [True, False])
def test_complex_periodic_vmc_trajectory_matches_reference(fused, monkeypatch):
    """vmc_worker with complex determinants (drift from Re grad, |ratio|^2 acceptance, complex block energies): fused
    wave-per-walker sweep and protocol path, replaying the reference's draws."""
    import pyqmc_amd as pa
    from helpers import pbc_complex_case
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g19_pbc_complex")
    sup, mf = pbc_complex_case()
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    cfg = PeriodicConfigs(g["vmc_start"].copy(), sup.lattice_vectors(), wrap=g["vmc_start_wrap"].copy())
    tstep, nsteps = float(g["vmc_tstep"]), int(g["vmc_nsteps"])
    acc = pa.EnergyAccumulator(sup, ewald_gmax=10)
    if fused:
        tapes = dict(gauss=g["vmc_gauss"], unif=g["vmc_unif"], ecp_rot=g["vmc_ecp_rot"], ecp_unif=g["vmc_ecp_unif"], record=[])
        blk, cfg = pa.vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc}, tapes=tapes)
        accepts = tapes["record"][0]
    else:
        gz, un = iter(g["vmc_gauss"].reshape(-1, *g["vmc_gauss"].shape[2:])), iter(g["vmc_unif"].reshape(-1, g["vmc_unif"].shape[-1]))
        monkeypatch.setattr(np.random, "normal", lambda scale, size: scale * next(gz))
        monkeypatch.setattr(np.random, "rand", lambda n: next(un))
        rots, eun = iter(g["vmc_ecp_rot"]), iter(g["vmc_ecp_unif"])
        accepts = []
        orig = wf.updateinternals
        monkeypatch.setattr(wf, "updateinternals", lambda e, ep, c, mask=None, saved_values=None: (accepts.append(mask.copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1])
        monkeypatch.setattr(acc, "avg", lambda c, w: {k: np.mean(v) for k, v in acc(c, w, rot=next(rots), unif=next(eun)).items()})
        blk, cfg = helpers.protocol_vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc})
        accepts = np.asarray(accepts).reshape(g["vmc_accepts"].shape)
    assert np.array_equal(np.asarray(accepts, dtype=bool), g["vmc_accepts"])
    assert helpers.relerr(cfg.configs, g["vmc_final"]) < 1e-9 and np.array_equal(cfg.wrap, g["vmc_final_wrap"])
    sign, logv = wf.value()
    assert helpers.relerr(logv, g["vmc_final_log"]) < 1e-9 and helpers.relerr(sign, g["vmc_final_sign"]) < 1e-8
    for k in ("energyke", "energyee", "energyei", "energyecp", "energygrad2", "energytotal", "acceptance"):
        assert helpers.relerr(blk[k], g[f"vmc_blk_{k}"]) < 1e-8, k


# ------------------------------------------------------------------ twisted boundary conditions
def _gpu_twisted_wf(tag):
    import pyqmc_amd as pa
    from helpers import twist_case

    sup, mf = twist_case(tag)
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    assert wf.fused_device().twisted and wf.wf_factors[0].dtype == complex
    return sup, wf


@pytest.mark.parametrize("tag", ["prim", "s211", "s222"])
def test_complex_lane_and_wave_per_walker_sweeps_agree(tag, monkeypatch):
    """Complex determinants (twisted cells): the lane-per-walker sweep (complex Sherman-Morrison on SoA planes, |ratio|^2
    acceptance, Re(grad) drift; default) and the wave-per-walker complex kernels follow the same Philox streams — same
    decisions, coordinates and energies to rounding, and both leave a state that equals a fresh recompute.  s222: 32 complex
    electrons per spin, the largest the lane-per-walker kernels take (64 doubles per inverse row, 80 KB of flush staging)."""
    import pyqmc_amd as pa

    res = []
    for lw in ("1", "0"):
        monkeypatch.setenv("PQA_LW", lw)
        sup, wf = _gpu_twisted_wf(tag)
        dev = wf.fused_device()
        cfg = pa.initial_guess(sup, 300, rng=np.random.default_rng(5))
        wf.recompute(cfg)
        acc, en, rec = dev.vmc_sweeps(0.3, 2, seed=77, energy=True, record=True)
        x = dev.configs()
        res.append((x, dev.value(), en, rec, dev.recompute(x), acc))
    same = res[0][3] == res[1][3]
    assert same.mean() > 0.9999
    ok = same.all(axis=(0, 1))
    assert ok.mean() > 0.98
    assert helpers.relerr(res[0][0][ok], res[1][0][ok]) < 1e-10
    assert np.max(np.abs(res[0][1][1][ok] - res[1][1][1][ok])) < 1e-9 and np.max(np.abs(res[0][1][0][ok] - res[1][1][0][ok])) < 1e-9
    assert helpers.relerr(res[0][2], res[1][2]) < 1e-3 and np.iscomplexobj(res[0][2])
    for r in res:  # updated phase and log|Psi| equal a fresh recompute
        assert np.max(np.abs(r[1][1] - r[4][1])) < 1e-8 and np.max(np.abs(r[1][0] - r[4][0])) < 1e-8


@pytest.mark.parametrize("tag", ["prim", "s211"])
def test_twisted_slater_matches_reference(tag):
    """Non-zero supercell twist (one twisted k-point in the primitive cell; two in a 2x1x1 supercell): complex
    lattice-summed AOs sum_L exp(i k_t.L) phi(r-R-L) on the device and the wrap phase of electrons that left the cell,
    derived from the unfolded positions the handle works with — MOs and the whole protocol against the reference."""
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g20_pbc_twist")
    sup, wf = _gpu_twisted_wf(tag)
    sl = wf.wf_factors[0]
    pts = PeriodicConfigs(g[f"{tag}_pts"].copy(), sup.lattice_vectors(), wrap=g[f"{tag}_pts_wrap"].copy())
    unfolded = (pts.configs + pts.wrap @ sup.lattice_vectors()).reshape(-1, 3)
    for nm, nc in (("val", 1), ("lap", 5)):
        ref = g[f"{tag}_mo_{nm}"]
        assert helpers.relerr(sl._dev.eval_mo(0, unfolded, nc), ref.reshape((nc, -1, ref.shape[-1]))) < 1e-12, nm
    err = run_protocol_pbc({"slater": sl, "jastrow": wf.wf_factors[1], "wf": wf}, g, f"{tag}_", sup)
    assert max(err.values()) < 2e-9, {k: v for k, v in err.items() if v > 1e-10}


@pytest.mark.parametrize("fused", [True, False])
def test_twisted_energy_and_vmc_match_reference(fused, monkeypatch):
    import pyqmc_amd as pa
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g20_pbc_twist")
    sup, wf = _gpu_twisted_wf("prim")
    cfg = PeriodicConfigs(g["en_configs"].copy(), sup.lattice_vectors(), wrap=g["en_wrap"].copy())
    wf.recompute(cfg)
    en = pa.EnergyAccumulator(sup, ewald_gmax=10)(cfg, wf, rot=g["en_rot"], unif=g["en_unif"])
    for k in ("ke", "ee", "ei", "ecp", "grad2", "total"):
        assert helpers.relerr(en[k], g[f"en_{k}"]) < 2e-9, k
    cfg = PeriodicConfigs(g["vmc_start"].copy(), sup.lattice_vectors(), wrap=g["vmc_start_wrap"].copy())
    tstep, nsteps = float(g["vmc_tstep"]), int(g["vmc_nsteps"])
    acc = pa.EnergyAccumulator(sup, ewald_gmax=10)
    if fused:
        tapes = dict(gauss=g["vmc_gauss"], unif=g["vmc_unif"], ecp_rot=g["vmc_ecp_rot"], ecp_unif=g["vmc_ecp_unif"], record=[])
        blk, cfg = pa.vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc}, tapes=tapes)
        accepts = tapes["record"][0]
    else:
        gz, un = iter(g["vmc_gauss"].reshape(-1, *g["vmc_gauss"].shape[2:])), iter(g["vmc_unif"].reshape(-1, g["vmc_unif"].shape[-1]))
        monkeypatch.setattr(np.random, "normal", lambda scale, size: scale * next(gz))
        monkeypatch.setattr(np.random, "rand", lambda n: next(un))
        rots, eun = iter(g["vmc_ecp_rot"]), iter(g["vmc_ecp_unif"])
        accepts = []
        orig = wf.updateinternals
        monkeypatch.setattr(wf, "updateinternals", lambda e, ep, c, mask=None, saved_values=None: (accepts.append(mask.copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1])
        monkeypatch.setattr(acc, "avg", lambda c, w: {k: np.mean(v) for k, v in acc(c, w, rot=next(rots), unif=next(eun)).items()})
        blk, cfg = helpers.protocol_vmc_worker(wf, cfg, tstep, nsteps, {"energy": acc})
        accepts = np.asarray(accepts).reshape(g["vmc_accepts"].shape)
    assert np.array_equal(np.asarray(accepts, dtype=bool), g["vmc_accepts"])
    assert helpers.relerr(cfg.configs, g["vmc_final"]) < 1e-9 and np.array_equal(cfg.wrap, g["vmc_final_wrap"])
    sign, logv = wf.value()
    assert helpers.relerr(logv, g["vmc_final_log"]) < 1e-9 and helpers.relerr(sign, g["vmc_final_sign"]) < 1e-8
    for k in ("energyke", "energyee", "energyei", "energyecp", "energygrad2", "energytotal", "acceptance"):
        assert helpers.relerr(blk[k], g[f"vmc_blk_{k}"]) < 1e-8, k


def test_periodic_obdm_orbitals_and_accumulator():
    """obdm.OrbitalEvaluator with k-points (the role of PBCOrbitalEvaluatorKpoints in obdm.py:85-91): the folded
    coefficient-only handle reproduces the oracle's Bloch orbitals at points inside and outside the cell, and the
    accumulator runs on a periodic wave function (auxiliary walkers folded by make_irreducible, testvalue_many over
    minimal images) with its invariants: norm >= 0 summing to norb, Hermitian-symmetric expectation within noise."""
    import pyqmc_amd as pa
    from oracle import pbc as opbc

    sup, wf = helpers.gpu_pbc_wf("fcc2cubic")
    _, kmf = helpers.pbc_slater_case("fcc2cubic")
    kpts = np.asarray(kmf.kpts)
    orb = [np.asarray(kmf.mo_coeff[0][k])[:, :3] for k in range(len(kpts))]
    ev = pa.obdm.OrbitalEvaluator(sup, orb, kpts=kpts)
    assert ev.norb == 3 * len(kpts) and ev.mo_dtype is float
    pts = (np.random.default_rng(4).random((40, 3)) * 3 - 1) @ sup.lattice_vectors()
    ref = opbc.PeriodicOrbitals(sup, kpts, [orb, orb], golden("g15_pbc_orbitals")["fcc2cubic_Ls"])
    assert helpers.relerr(ev.mos(pts), ref.mos(ref.aos(pts, 1), 0)[0]) < 1e-9
    np.random.seed(9)
    cfg = pa.initial_guess(sup, 24, rng=np.random.default_rng(2))
    wf.recompute(cfg)
    acc = pa.OBDMAccumulator(sup, orb, kpts=kpts, nsweeps=2, warmup=4, spin=0)
    d = acc(cfg, wf)
    assert d["value"].shape == (24, ev.norb, ev.norb) and d["norm"].shape == (24, ev.norb)
    assert np.all(d["norm"] >= 0) and np.allclose(d["norm"].sum(axis=1), ev.norb) and np.all(np.isfinite(d["value"]))
    frac = acc._extra_config.configs @ np.linalg.inv(sup.lattice_vectors())
    assert frac.min() >= -1e-12 and frac.max() < 1 + 1e-12  # the auxiliary walkers live in the cell


def test_twisted_obdm_is_invariant_under_lattice_translations_of_the_electrons():
    """Twisted cell (complex Bloch orbitals, s211 of g20): the estimator multiplies conj(Psi(R')/Psi(R)) by
    conj(phi_k(r_e)); translating electron e by a lattice vector L multiplies Psi(R) by exp(i k.L) and phi_k(r_e) by the
    same phase (orbitals.py:201-213), so the density matrix must not change.  That only holds when the basis orbitals at
    the electrons carry the wrap phase: the same folded configurations with wrap counters 0 and with non-zero wrap counters
    (same physical state) must give the same values; so must the true (unfolded) coordinates handed over directly."""
    import pyqmc_amd as pa
    from helpers import twist_case
    from pyqmc_amd.configs import PeriodicConfigs

    sup, wf = _gpu_twisted_wf("s211")
    _, kmf = twist_case("s211")
    kpts = np.asarray(kmf.kpts)
    orb = [np.asarray(kmf.mo_coeff[0][k])[:, :2] for k in range(len(kpts))]
    lat = sup.lattice_vectors()
    base = pa.initial_guess(sup, 12, rng=np.random.default_rng(3))
    wrap = np.random.default_rng(4).integers(-2, 3, size=base.configs.shape).astype(float)
    out = []
    for w in (np.zeros_like(wrap), wrap):
        cfg = PeriodicConfigs(base.configs.copy(), lat, wrap=w.copy())
        assert np.allclose(cfg.configs, base.configs, atol=1e-12) and np.array_equal(cfg.wrap, w)  # same folded positions, other wrap counters
        wf.recompute(cfg)
        acc = pa.OBDMAccumulator(sup, orb, kpts=kpts, nsweeps=2, warmup=3, tstep=0.4)
        assert acc.dtype is complex
        np.random.seed(21)
        out.append(acc(cfg, wf))
    assert np.iscomplexobj(out[0]["value"]) and np.abs(out[0]["value"].imag).max() > 1e-6
    assert helpers.relerr(out[1]["value"], out[0]["value"]) < 1e-9 and helpers.relerr(out[1]["norm"], out[0]["norm"]) < 1e-12
    # the auxiliary walk in unfolded coordinates reports walkers inside the cell with their wrap counters
    aux = acc._extra_config
    frac = aux.configs @ np.linalg.inv(lat)
    assert frac.min() >= -1e-12 and frac.max() < 1 + 1e-12 and np.abs(aux.wrap).max() >= 0


def test_periodic_orbital_tile_widths_are_bitwise_identical(monkeypatch):
    """The periodic k_orb picks its point-tile width (16 for small launches, 32 / 64 by timing both on large ones); that is only legitimate
    because the two instantiations produce the same bits (same chunk composition => same MFMA accumulation order)."""
    import pyqmc_amd as pa

    sup, mf = helpers.pbc_slater_case("fcc2cubic")
    pts = (np.random.default_rng(8).random((700, 3)) * 3 - 1) @ sup.lattice_vectors()
    rows = {}
    for tp in ("16", "32", "64"):
        monkeypatch.setenv("PQA_ORB_TP", tp)
        dev = pa.generate_wf(sup, mf).fused_device()
        rows[tp] = [dev.eval_mo(0, pts, nc) for nc in (1, 5)]
    for other in ("16", "64"):
        for a, b in zip(rows["32"], rows[other]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("tag", ["fcc2cubic", "k222", "twist_s211"])
def test_periodic_image_lists_and_direct_tests_agree(tag, monkeypatch):
    """The lattice sums walk per-(point, atom) image lists written by the pre-pass (ordered by shell range, nearest image
    first).  A lane whose list does not fit its PQA_PBC_NW words tests every candidate image directly instead — the path
    the thread-per-point AO kernel (eval_ao / use_mfma=False) always takes.  Forcing one-word lists (3 entries) sends
    nearly every lane down the direct path: orbitals from list walks, from direct tests and from the AO kernel must agree
    to rounding (the order of the image sum differs), for points inside and well outside the cell."""
    import pyqmc_amd as pa

    sup, mf = helpers.twist_case("s211") if tag == "twist_s211" else helpers.pbc_slater_case(tag)
    pts = (np.random.default_rng(11).random((1500, 3)) * 4 - 1.5) @ sup.lattice_vectors()
    rows = {}
    for nw in ("", "1"):
        if nw:
            monkeypatch.setenv("PQA_PBC_NW", nw)
        dev = pa.generate_wf(sup, mf).fused_device()
        rows[nw] = [dev.eval_mo(0, pts, nc) for nc in (1, 5)] + ([dev.eval_mo(0, pts, 5, use_mfma=False)] if tag != "twist_s211" else [])
    for a, b in zip(rows[""], rows["1"]):
        assert helpers.relerr(a, b) < 1e-13
    if tag != "twist_s211":  # (the thread-per-point AO kernel is real-only: no twisted lattice sums)
        assert helpers.relerr(rows[""][1], rows[""][2]) < 1e-13


@pytest.mark.parametrize("tag", ["fcc2cubic", "k222", "twist_s211", "general"])
def test_near_candidate_masks_do_not_change_a_bit(tag, monkeypatch):
    """The pre-pass only distance-tests the candidate images a table marks as reachable from the sub-cell the folded displacement
    point - atom falls in (create: near_masks; PQA_PRE_GRID sub-cells per axis, 0 = test every candidate).  The table is a superset
    of what the exact tests admit, so the image lists — and with them every orbital value — must be the SAME BITS with it,
    without it and with a coarse grid; points on sub-cell faces, on cell faces and far outside the cell included."""
    import pyqmc_amd as pa

    if tag == "general":  # a sheared cell: the sub-cells are parallelepipeds whose half diagonal is not a coordinate axis
        sup, mf = helpers.pbc_slater_case("gamma")
    else:
        sup, mf = helpers.twist_case("s211") if tag == "twist_s211" else helpers.pbc_slater_case(tag)
    lat = sup.lattice_vectors()
    rng = np.random.default_rng(12)
    frac = rng.random((3000, 3)) * 5 - 2
    frac[:600] = np.rint(frac[:600] * 16) / 16          # exactly on the faces of the 8^3 and 16^3 grids (and on cell faces)
    frac[600:900, 0] = 0.5; frac[900:1200, 1] = -0.5     # the fold's own boundary
    pts = frac @ lat
    atoms = np.asarray(sup.atom_coords())
    pts[1200:1200 + len(atoms)] = atoms                 # on the nuclei: folded displacement exactly 0
    rows = {}
    for g in ("", "0", "3", "16", "ncut10"):
        if g == "ncut10":  # the pre-pass instantiation for atoms with more than five distinct shell cut-offs
            monkeypatch.delenv("PQA_PRE_GRID")
            monkeypatch.setenv("PQA_PRE_NCUT", "10")
        elif g:
            monkeypatch.setenv("PQA_PRE_GRID", g)
        dev = pa.generate_wf(sup, mf).fused_device()
        rows[g] = [dev.eval_mo(0, pts, nc) for nc in (1, 5)]
    for g in ("0", "3", "16", "ncut10"):
        for a, b in zip(rows[""], rows[g]):
            assert np.array_equal(a, b), g


@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic"])
def test_periodic_pgradient_matches_reference(tag):
    """pgradient() of a periodic Slater-Jastrow (slater.py:462-542, orbitals.py:239-254): the orbital coefficients are
    exposed in the reference's layout — per-k blocks (nao_prim, nmo_k) concatenated over k — folded into supercell
    coefficients when pushed, and the device gradient (w.r.t. the folded matrix) is mapped back by the chain rule."""
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g24_pbc_pgrad")
    sup, wf = helpers.gpu_pbc_wf(tag)
    for k in ("wf1det_coeff", "wf1mo_coeff_alpha", "wf1mo_coeff_beta", "wf2acoeff", "wf2bcoeff"):
        assert np.shape(wf.parameters[k]) == g[f"{tag}_param_{k}"].shape, k
        assert helpers.relerr(wf.parameters[k], g[f"{tag}_param_{k}"]) < 1e-13, k
    cfg = PeriodicConfigs(g[tag + "_configs"].copy(), sup.lattice_vectors())
    s0, l0 = wf.recompute(cfg)
    pg = wf.pgradient()
    assert sorted(pg.keys()) == g[tag + "_keys"].tolist()
    for k, v in pg.items():
        assert v.shape == g[f"{tag}_pgrad_{k}"].shape and helpers.relerr(v, g[f"{tag}_pgrad_{k}"]) < 1e-8, k
    # assigning in the reference's layout reaches the device: same values give the same wave function, a finite
    # difference along one coefficient reproduces its logarithmic derivative
    wf.parameters["wf1mo_coeff_alpha"] = g[f"{tag}_param_wf1mo_coeff_alpha"].copy()
    s1, l1 = wf.recompute(cfg)
    assert np.array_equal(s0, s1) and np.max(np.abs(l0 - l1)) < 1e-12
    C = g[f"{tag}_param_wf1mo_coeff_alpha"].copy()
    mu, col, h = 5, C.shape[1] - 1, 1e-5
    vals = []
    for sgn in (1, -1):
        Cp = C.copy()
        Cp[mu, col] += sgn * h
        wf.parameters["wf1mo_coeff_alpha"] = Cp
        vals.append(wf.recompute(cfg)[1])
    wf.parameters["wf1mo_coeff_alpha"] = C
    assert np.allclose((vals[0] - vals[1]) / (2 * h), pg["wf1mo_coeff_alpha"][:, mu, col], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["cplx", "twist"])
def test_complex_testvalue_many_matches_reference(tag):
    """testvalue_many with complex determinants (k_testvalue_many<true>: complex ratio dots on [Re | Im] orbital rows, real
    Jastrow factor): complex Bloch coefficients at zero twist and a twisted 2x1x1 cell, auxiliary positions inside and
    outside the cell (the handle derives the moved electron's wrap phase from the unfolded position)."""
    import pyqmc_amd as pa
    from helpers import pbc_complex_case
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g25_complex_testvalue_many")
    if tag == "cplx":
        sup, mf = pbc_complex_case()
        wf = pa.generate_wf(sup, mf)
        wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = pbc_jastrow_coeffs(sup)
    else:
        sup, wf = _gpu_twisted_wf("s211")
    cfg = PeriodicConfigs(g[tag + "_configs"].copy(), sup.lattice_vectors(), wrap=g[tag + "_wrap"].copy())
    wf.recompute(cfg)
    epos = cfg.make_irreducible(0, g[tag + "_aux"])
    es = g[tag + "_es"]
    for nm, w in (("slater", wf.wf_factors[0]), ("j2", wf.wf_factors[1]), ("wf", wf)):
        got = w.testvalue_many(es, epos)
        assert got.dtype == g[f"{tag}_{nm}"].dtype and got.shape == g[f"{tag}_{nm}"].shape, nm
        assert helpers.relerr(got, g[f"{tag}_{nm}"]) < 2e-9, nm
    # column i equals testvalue(es[i]); a mask returns the masked rows
    assert helpers.relerr(wf.testvalue_many(np.array([int(es[1])]), epos)[:, 0], wf.testvalue(int(es[1]), epos)[0]) < 1e-12
    mask = np.array([True, False, True])
    assert np.array_equal(wf.testvalue_many(es, epos, mask=mask), wf.testvalue_many(es, epos)[mask])


@pytest.mark.parametrize("periodic,resident", [(False, False), (False, True), (True, False)])
def test_event_brackets_do_not_change_the_sweep(periodic, resident, monkeypatch):
    """The measurement entry points (pqa_profile_enable / pqa_profile_query*: HIP events around a sample of the orbital,
    partial-sum and flush launches, read by bench.py and tools/pbc_bench.py) must leave the numbers alone: the same seeded
    sweep with and without them gives identical acceptances and energies, and the queries report bracketed launches with a
    positive duration on the open and on the periodic (pre-pass + lattice-sum) path."""
    import pyqmc_amd as pa

    monkeypatch.delenv("PQA_LW", raising=False)  # the partial-sum brackets belong to the (default) lane-per-walker sweep
    monkeypatch.setenv("PQA_RES", "1" if resident else "0")  # resident sweep: ONE bracketed launch per sweep (W x N x 5 point-components)
    if periodic:
        sup, mf = helpers.pbc_slater_case("fcc2cubic")
    else:
        sup = systems.water()
        mf = systems.random_mf(sup)
    out = {}
    for prof in (False, True):
        wf = pa.generate_wf(sup, mf)
        dev = wf.fused_device()
        wf.recompute(pa.initial_guess(sup, 256, rng=np.random.default_rng(3)))
        dev.profile_enable(prof)
        acc, en, _ = dev.vmc_sweeps(0.3, 2, seed=9, energy=True)
        dev.sync()
        if prof:
            launches, ms, point_comps = dev.profile_query()
            if resident:
                assert launches == 2 and ms > 0.0 and point_comps == launches * 256 * dev.N * 5
            else:
                assert launches > 0 and ms > 0.0 and point_comps == launches * 256 * 5
                p_launches, p_ms, groups = dev.profile_query_part()
                assert p_launches > 0 and p_ms > 0.0 and groups >= 1
            dev.profile_enable(False)
        out[prof] = (np.array(acc), np.array(en))
    assert np.array_equal(out[False][0], out[True][0]) and np.array_equal(out[False][1], out[True][1])


@pytest.mark.gpu
def test_complex_ecp_points_agree_with_the_wave_per_walker_accumulation(monkeypatch):
    """Twisted cell, fused sweep energies: the thread-per-point ECP kernel on the complex planes (k_ecp_point_lw<.., CX>, default)
    against the wave-per-walker accumulation it replaces there (PQA_ECP_POINT_LW=0): same points, complex sums in another order."""
    import pyqmc_amd as pa
    from pyqmc_amd import pbc, systems

    sup = pbc.get_supercell(systems.diamond_primitive(), np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]))
    mf = pbc.random_kmf(sup, complex_coeff=True, twist=(0.25, 0.1, -0.3))
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("PQA_ECP_POINT_LW", flag)
        wf = pa.generate_wf(sup, mf)
        wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = helpers.pbc_jastrow_coeffs(sup)
        dev = wf.fused_device()
        wf.recompute(pa.initial_guess(sup, 500, rng=np.random.default_rng(2)))
        acc, en, _ = dev.vmc_sweeps(0.3, 2, seed=19, energy=True)
        out.append((dev.configs(), np.asarray(en)))
    assert np.array_equal(out[0][0], out[1][0])
    assert np.max(np.abs(out[0][1][:, 3])) > 0 and np.max(np.abs(out[0][1].imag)) > 0  # there is an ECP term, with an imaginary part
    assert note("complex_ecp_point_vs_accum", np.max(np.abs(out[0][1] - out[1][1]) / np.maximum(1.0, np.abs(out[1][1])))) < 1e-12


@pytest.mark.gpu
def test_fold_only_minimal_image_in_the_jastrow_pairs_is_bitwise_the_full_reduction(monkeypatch):
    """General (fcc-type) cell with the Jastrow cut-offs at the inradius of the fold's parallelepiped (pyqmc's periodic default):
    the pair loops skip the Voronoi reduction (min_image_j).  Inside the cut-off the folded vector is the minimal image and
    outside it the pair does not contribute either way, so walkers, log-values, energies and DMC weights must equal the
    PQA_JAS_FOLD=0 run bit for bit."""
    import pyqmc_amd as pa
    from pyqmc_amd import pbc, systems

    sup = pbc.get_supercell(systems.diamond_primitive(), 2.0 * np.eye(3))
    mf = pbc.random_kmf(sup)
    out = []
    for flag in ("0", "1"):
        monkeypatch.setenv("PQA_JAS_FOLD", flag)
        wf = pa.generate_wf(sup, mf)
        wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = helpers.pbc_jastrow_coeffs(sup)
        dev = wf.fused_device()
        wf.recompute(pa.initial_guess(sup, 600, rng=np.random.default_rng(12)))
        acc, en, _ = dev.vmc_sweeps(0.3, 2, seed=23, energy=True)
        w = np.ones(600)
        et = float(np.real(en[-1][5]))
        avg, dacc = dev.dmc_steps(0.02, 2, w, 10.0, et, et, seed=5)
        out.append((dev.configs(), dev.value()[1], np.asarray(en), avg.copy(), dacc.copy(), w.copy()))
    for p, q in zip(*out):
        assert np.array_equal(p, q)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["cplx", "twist_prim", "twist_s211"])
def test_complex_pgradient_matches_reference(tag):
    """Slater.pgradient for COMPLEX determinants (slater.py:462-542 in complex arithmetic): complex Bloch phases
    and coefficients on the 3x1x1 supercell, and twisted cells whose AOs are complex lattice sums (k_ao_tw) — walkers partly
    outside the cell, so every electron's wrap phase enters.  The derivatives w.r.t. the determinant coefficients and the
    per-k orbital blocks (the reference's parameter layout: complex (nao_prim, sum_k nmo_k)) against the reference's
    (tests/golden/g31_complex_pgrad.npz), for the bare Slater factor and inside the product; plus a complex finite difference
    along one coefficient (holomorphic: the same derivative along the real and the imaginary direction)."""
    import pyqmc_amd as pa
    from helpers import pbc_complex_case, twist_case
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g31_complex_pgrad")
    sup, mf = pbc_complex_case() if tag == "cplx" else twist_case(tag[6:])
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    sl = wf.wf_factors[0]
    for k in ("det_coeff", "mo_coeff_alpha", "mo_coeff_beta"):  # the parameters themselves, in the reference's per-k layout
        ref = g[f"{tag}_param_{k}"]
        assert np.shape(sl.parameters[k]) == ref.shape and helpers.relerr(sl.parameters[k], ref) < 1e-13, k
    cfg = PeriodicConfigs(g[tag + "_configs"].copy(), sup.lattice_vectors(), wrap=g[tag + "_wrap"].copy())
    s0, l0 = wf.recompute(cfg)
    for nm, w, pre in (("slater", sl, ""), ("wf", wf, "wf1")):
        pg = w.pgradient()
        assert sorted(pg.keys()) == g[f"{tag}_{nm}_keys"].tolist()
        for k, v in pg.items():
            ref = g[f"{tag}_{nm}_pgrad_{k}"]
            assert v.shape == ref.shape and v.dtype == ref.dtype, k
            assert note(f"cpgrad_{tag}_{nm}_{k}", helpers.relerr(v, ref)) < 1e-8, k
    pg = sl.pgradient()
    C = np.array(sl.parameters["mo_coeff_beta"])
    mu, col, h = 7, C.shape[1] - 1, 1e-5
    for direction in (1.0, 1.0j):
        vals = []
        for sgn in (1, -1):
            Cp = C.copy()
            Cp[mu, col] += sgn * h * direction
            sl.parameters["mo_coeff_beta"] = Cp
            ph, lg = sl.recompute(cfg)
            vals.append(np.log(ph) + lg)
        fd = (vals[0] - vals[1]) / (2 * h * direction)
        fd = fd.real + 1j * ((fd.imag + np.pi) % (2 * np.pi) - np.pi)
        assert np.allclose(fd, pg["mo_coeff_beta"][:, mu, col], rtol=5e-5, atol=2e-6), direction
    sl.parameters["mo_coeff_beta"] = C


@pytest.mark.gpu
def test_stochastic_reconfiguration_on_a_twisted_cell():
    """BASELINE config C3's wave function (8-atom diamond cell, k-point twist: complex determinants) through
    LinearTransform + StochasticReconfiguration: complex per-k orbital blocks serialise to real + imaginary parameters
    (accumulators.py:134-150), the moments <dp f dp^T>, <E f dp>, <f dp> come from the device product (pqa_gram on the real and
    imaginary parts) and equal the reference's einsum formulas on the same derivatives; an SR step is finite."""
    import pyqmc_amd as pa

    sup = pbc.get_supercell(systems.diamond_primitive(), np.array([[-1.0, 1, 1], [1, -1, 1], [1, 1, -1]]))
    mf = pbc.random_kmf(sup, complex_coeff=True, twist=(0.25, 0.1, -0.3), nvirt=1)
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    assert wf.fused_device().twisted and np.iscomplexobj(wf.parameters["wf1mo_coeff_alpha"])
    rng = np.random.default_rng(3)
    to_opt = {"wf1mo_coeff_alpha": rng.random(np.shape(wf.parameters["wf1mo_coeff_alpha"])) < 0.05, "wf2bcoeff": np.ones((4, 3), dtype=bool)}
    to_opt["wf2bcoeff"][0] = False
    tr = pa.LinearTransform(wf.parameters, to_opt)
    nre = int(to_opt["wf1mo_coeff_alpha"].sum()) + 9
    assert tr.nparams == nre and len(tr.serialize_parameters(wf.parameters)) == nre + int(to_opt["wf1mo_coeff_alpha"].sum())
    class OneEnergy:  # the ECP quadrature draws fresh rotations per evaluation: both routes below must see the same energies
        def __init__(self, acc):
            self.acc, self.last = acc, None

        def __call__(self, configs, wf_):
            if self.last is None:
                self.last = self.acc(configs, wf_)
            return {k: v.copy() for k, v in self.last.items()}

    sr = pa.StochasticReconfiguration(OneEnergy(pa.EnergyAccumulator(sup, ewald_gmax=10)), tr)
    cfg = pa.initial_guess(sup, 64, rng=np.random.default_rng(5))
    wf.recompute(cfg)
    d = sr.avg(cfg, wf)
    dp, en, fdp = sr._derivatives(cfg, wf, 1e-3)
    w = np.full(64, 1.0 / 64)
    assert np.iscomplexobj(dp) and dp.shape == (64, len(tr.serialize_parameters(wf.parameters)))
    assert helpers.relerr(d["dpidpj"], np.einsum("ij,ik->jk", dp, w[:, None] * fdp)) < 1e-12
    assert helpers.relerr(d["dpH"], np.einsum("i,ij->j", en["total"], w[:, None] * fdp)) < 1e-12
    assert helpers.relerr(d["dppsi"], np.average(fdp, weights=w, axis=0)) < 1e-12
    steps, rep = sr.delta_p([0.1], d)
    assert np.all(np.isfinite(steps[0])) and steps[0].shape == (len(tr.serialize_parameters(wf.parameters)),)
    new = tr.deserialize(wf, tr.serialize_parameters(wf.parameters) + steps[0])
    for k, v in new.items():
        wf.parameters[k] = v
    assert np.all(np.isfinite(wf.recompute(cfg)[1]))


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["gamma", "fcc2cubic"])
def test_periodic_g_and_h_shells_match_reference(tag):
    """Periodic cells with l = 4, 5 shells (the reference's pbcgto.py goes to SPH5): the MFMA orbital kernels' lattice-sum phase is
    built for l <= 3, so such handles evaluate their orbitals on the general path — thread-per-point AOs with the same image
    tests (k_ao<.., 5>), contracted by k_mo_rows.  AOs, MOs and the whole Slater-Jastrow protocol against the reference (g32),
    then a fused VMC sweep + energy (ECP points and all go through the same path) against the oracle on the same tapes."""
    import pyqmc_amd as pa
    from helpers import PBC_SLATER_CASES, unfold_ao
    from oracle import jastrow_basis, vmc as ovmc, wf as owf
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g32_pbc_high_l")
    sup = pbc.get_supercell(systems.diamond_primitive_high_l(), PBC_SLATER_CASES[tag])
    mf = pbc.random_kmf(sup)
    wf = pa.generate_wf(sup, mf)
    a, b = pbc_jastrow_coeffs(sup)
    wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
    sl = wf.wf_factors[0]
    pts = g[f"{tag}_pts"].reshape(-1, 3)
    for nm, nc in (("val", 1), ("grad", 4), ("lap", 5)):
        ao = sl._dev.eval_ao(pts, nc)
        ref = g[f"{tag}_ao_{nm}"]
        assert note(f"pbc_high_l_{tag}_ao_{nm}", helpers.relerr(unfold_ao(sup, mf.kpts, ao), ref.reshape((ref.shape[0], nc, -1, ref.shape[-1])))) < 1e-12
    for nm, nc in (("val", 1), ("lap", 5)):
        ref = g[f"{tag}_mo_{nm}"]
        assert helpers.relerr(sl._dev.eval_mo(0, pts, nc), ref.reshape((nc, -1, ref.shape[-1]))) < 1e-12, nm
    err = run_protocol_pbc({"slater": sl, "jastrow": wf.wf_factors[1], "wf": wf}, g, f"{tag}_", sup)
    assert max(err.values()) < 2e-9, {k: v for k, v in err.items() if v > 1e-10}
    # fused sweep + energy against the oracle, replayed tapes
    W, N, necp = 6, int(sum(sup.nelec)), sup.natm
    rng = np.random.default_rng(8)
    start = pa.initial_guess(sup, W, rng=rng)
    gauss, unif = rng.standard_normal((1, N, W, 3)), rng.random((1, N, W))
    rot = np.broadcast_to(np.eye(3), (1, N, necp, 3, 3)).copy()
    eunif = rng.random((1, N, necp, W))
    blk, cfg = pa.vmc_worker(wf, PeriodicConfigs(start.configs.copy(), sup.lattice_vectors()), 0.3, 1, {"energy": pa.EnergyAccumulator(sup, ewald_gmax=10)},
                             tapes=dict(gauss=gauss, unif=unif, ecp_rot=rot, ecp_unif=eunif))
    Ls = pbc.lattice_points_within(sup.original_cell.lattice_vectors(), 30.0 + pbc.cell_diameter(sup.original_cell.lattice_vectors()))
    osl = owf.Slater.periodic(sup, mf.kpts, mf.mo_coeff, Ls)
    rcut = float(np.amin(np.pi / np.linalg.norm(sup.reciprocal_vectors(), axis=1)))
    ab, bb, rcut = jastrow_basis.default_basis(ion_cusp=False, rcut=rcut)
    oja = owf.JastrowSpin(sup, ab, bb, rcut)
    oja.parameters["acoeff"], oja.parameters["bcoeff"] = a, b
    oblk, ocfg = ovmc.vmc_worker(sup, owf.MultiplyWF(osl, oja), PeriodicConfigs(start.configs.copy(), sup.lattice_vectors()), 0.3, gauss, unif, rot, eunif,
                                 ewald_kws={"ewald_gmax": 10})
    assert abs(blk["acceptance"] - oblk["acceptance"]) < 1e-12
    assert note(f"pbc_high_l_{tag}_sweep_dx", np.max(np.abs(cfg.configs - ocfg.configs))) < 1e-9
    assert abs(blk["energytotal"] - oblk["energytotal"]) < 1e-8 * abs(oblk["energytotal"])
