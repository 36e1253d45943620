"""GPU parity of the periodic-boundary pieces of the hot path against outputs of the real reference
(tests/golden/g14_pbc_jastrow.npz; generator tests/golden/make_golden.py:g_pbc).  All calls go through the C ABI."""

import numpy as np
import pytest

import helpers
from helpers import PBC_JASTROW_CASES, golden, pbc_jastrow_coeffs, run_protocol_pbc
from pyqmc_amd import systems

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


def test_periodic_entry_points_fail_loudly_until_implemented():
    """No silent open-boundary answer for a periodic system."""
    import pyqmc_amd as pa
    from pyqmc_amd import _ffi

    cell = systems.diamond_primitive()
    ja, _ = pa.wf.generate_jastrow(cell)
    ja.recompute(systems.initial_guess(cell, 4))
    with pytest.raises(_ffi.PqaError):
        ja._dev.energy(10.0)


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
