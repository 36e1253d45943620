"""CPU oracle for the trial-wave-function hot path — TEST INFRASTRUCTURE, NOT PRODUCT.

A NumPy restatement of the reference algorithm (WagnerGroup/pyqmc 0.8.0) for the
path named in BASELINE.json: GTO atomic orbitals, AO->MO contraction,
Slater determinant (recompute / Sherman-Morrison / ratios, gradients, Laplacians),
two-body Jastrow, product wave function, kinetic + Coulomb + ECP local energy and
the single-electron-move VMC sweep.  Every function cites the reference
file:line it follows.

Pinning: the restatement is checked in ``tests/test_oracle_golden.py`` against
golden vectors in ``tests/golden/*.npz`` that were produced by importing the real
reference (``tests/golden/make_golden.py``; pyscf/h5py mocked, numba replaced by an
identity decorator so the in-repo evaluators run as IEEE fp64 Python).  The AO
specification is the reference's in-repo evaluator ``pyqmc/wf/numba/gto.py``; its
agreement with PySCF's libcgto (un-vendored dependency ``pyscf>=2.8,<3``, the
reference's *default* AO backend) is pinned by the reference itself only to
3e-5 (``tests/unit/test_gto.py:114-134``) and is otherwise "parity unpinned".

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package — as the checker / reported baseline, never as a
compute path of ``pyqmc_amd``.
"""
