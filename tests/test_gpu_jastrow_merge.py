"""The merged evaluation of a Jastrow basis' PolyPade functions (csrc/pqa_jastrow.hpp: pade_merged — one rational function of p
per pair, numerator tables from pqa_capi.hip: jas_merge_tables) against the function-by-function route it replaces on the
lane-per-walker kernels (PQA_JAS_MERGE=0): jastrowspin.py:296-385 with func3d.py:25-49 summed in a different order, so the two
agree to rounding.  Both are pinned to the oracle elsewhere (tests/test_gpu_parity.py, test_gpu_fullsize.py)."""

import numpy as np
import pytest

from pyqmc_amd import systems
from pyqmc_amd.configs import OpenConfigs

from . import helpers
from .helpers import relerr

pytestmark = pytest.mark.gpu


def _sweep(wf, cfg, nsteps=2):
    dev = wf.fused_device()
    wf.recompute(cfg)
    acc, en, rec = dev.vmc_sweeps(0.3, nsteps, seed=77, energy=True, record=True)
    return dict(x=dev.configs(), log=dev.value()[1], en=np.asarray(en), rec=np.asarray(rec), acc=np.asarray(acc))


def _compare(a, b, tol_x=1e-10, tol_e=1e-9):
    same = a["rec"] == b["rec"]
    assert same.mean() > 0.9999  # a decision can only flip on a ~1e-13 near-tie
    ok = same.all(axis=tuple(range(same.ndim - 1)))
    assert ok.mean() > 0.98
    assert relerr(a["x"][ok], b["x"][ok]) < tol_x
    assert np.max(np.abs(a["log"][ok] - b["log"][ok])) < 1e-9
    if ok.all():
        assert relerr(a["en"], b["en"]) < tol_e


@pytest.mark.parametrize("ion_cusp", [False, True])
def test_merged_pade_sums_agree_with_the_function_by_function_route(ion_cusp, monkeypatch):
    """64-electron cluster, 300 walkers, the large-shard step kernel forced (k_step_lw with a wave per thread group: the route of
    the headline bench) + the kinetic-energy kernel: same decisions, coordinates, log|Psi| and energies with the tables on / off;
    then new coefficients through the parameter dictionary (the tables are rebuilt) and the same again.  ion_cusp: a cusp
    function ahead of four Pade functions in the electron-ion basis."""
    import pyqmc_amd as pa

    monkeypatch.setenv("PQA_STEP_PRE", "0")
    monkeypatch.setenv("PQA_LW_GM", "4")
    mol = systems.water_cluster()
    mf = systems.random_mf(mol)
    start = pa.initial_guess(mol, 300, rng=np.random.default_rng(5)).configs
    rng = np.random.default_rng(3)
    res = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("PQA_JAS_MERGE", merge)
        if ion_cusp:
            wf = pa.generate_wf(mol, mf, jastrow_kws=dict(ion_cusp=["O"]))
            a0 = np.array(wf.parameters["wf2acoeff"])
            a = 0.05 * np.random.default_rng(11).standard_normal(a0.shape)
            a[:, 0, :] = a0[:, 0, :]
            b = helpers.jastrow_params(mol)[1]
            wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, b
        else:
            wf = helpers.gpu_wf(mol, mf)
        first = _sweep(wf, OpenConfigs(start.copy()))
        a2 = np.array(wf.parameters["wf2acoeff"])
        b2 = np.array(wf.parameters["wf2bcoeff"])
        r2 = np.random.default_rng(21)
        a2[:, -3:, :] += 0.03 * r2.standard_normal(a2[:, -3:, :].shape)
        b2[1:] += 0.03 * r2.standard_normal(b2[1:].shape)
        wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a2, b2
        second = _sweep(wf, OpenConfigs(start.copy()))
        res[merge] = (first, second)
    assert res["1"][0]["acc"].mean() > 0.3
    assert np.abs(res["1"][0]["en"] - res["1"][1]["en"]).max() > 1e-6  # the new coefficients did change the energies
    _compare(res["1"][0], res["0"][0])
    _compare(res["1"][1], res["0"][1])
    # the two routes are different arithmetic: were the results bit-identical, the tables would not have been used
    assert not np.array_equal(res["1"][0]["x"], res["0"][0]["x"]) and not np.array_equal(res["1"][0]["en"], res["0"][0]["en"])


def test_merged_pade_sums_in_a_periodic_cell(monkeypatch):
    """2x2x2 diamond supercell (minimal images in the pair loops, ECP, Ewald): tables on / off."""
    import pyqmc_amd as pa

    monkeypatch.setenv("PQA_STEP_PRE", "0")
    monkeypatch.setenv("PQA_LW_GM", "4")
    sup, mf = helpers.pbc_slater_case("k222")
    start = pa.initial_guess(sup, 200, rng=np.random.default_rng(5))
    res = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("PQA_JAS_MERGE", merge)
        _, wf = helpers.gpu_pbc_wf("k222")
        res[merge] = _sweep(wf, start.copy(), nsteps=1)
    _compare(res["1"], res["0"], tol_e=1e-8)
    assert not np.array_equal(res["1"]["x"], res["0"]["x"])


def test_ion_cusp_basis_on_the_small_shard_and_ecp_routes(monkeypatch):
    """Five functions in the electron-ion basis (the ion cusp of all-electron atoms + four Pade functions) are more than the
    register-table route of the pair loops takes: where electron or partner index differs between lanes (the narrow step kernel
    small shards fall back to, the ECP list pass) the merged tables are read per lane instead of walking the function tables in
    the innermost loop.  Default kernel selection at 300 walkers, tables on / off."""
    import pyqmc_amd as pa

    mol = systems.water_cluster()
    mf = systems.random_mf(mol)
    start = pa.initial_guess(mol, 300, rng=np.random.default_rng(5)).configs
    res = {}
    for merge in ("1", "0"):
        monkeypatch.setenv("PQA_JAS_MERGE", merge)
        wf = pa.generate_wf(mol, mf, jastrow_kws=dict(ion_cusp=["O"]))
        a0 = np.array(wf.parameters["wf2acoeff"])
        a = 0.05 * np.random.default_rng(11).standard_normal(a0.shape)
        a[:, 0, :] = a0[:, 0, :]
        wf.parameters["wf2acoeff"], wf.parameters["wf2bcoeff"] = a, helpers.jastrow_params(mol)[1]
        res[merge] = _sweep(wf, OpenConfigs(start.copy()))
    _compare(res["1"], res["0"])
    assert not np.array_equal(res["1"]["x"], res["0"]["x"])
