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
