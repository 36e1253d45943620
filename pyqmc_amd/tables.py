"""Flatten a molecule / mean-field description into the arrays of ``pqa_system_t``.

Host-side set-up only (runs once per wave function): basis normalisation as the
reference's ``normalize_basis_coeffs`` (``pyqmc/wf/numba/gto.py:375-405``, PySCF
``gto_norm`` convention), AO ordering as ``AtomicOrbitalEvaluator.__init__``
(:435-470), determinant packing as ``determinant_tools.create_packed_objects``
(``pyqmc/wf/determinant_tools.py:39-71``), MO truncation as
``orbital_evaluator_from_pyscf`` (``pyqmc/pyscftools.py:181-183``), ECP channel
tables as ``generate_ecp_functors`` (``pyqmc/observables/eval_ecp.py:160-179``).
"""

import math

import numpy as np


def _shell_norm(l, exps, coefs):
    m = l + 1.5
    g = math.gamma(m)
    scaled = [c * math.sqrt(2.0 * (2.0 * a) ** m / g) for a, c in zip(exps, coefs)]
    s = 0.0
    for ap, cp in zip(exps, scaled):
        for aq, cq in zip(exps, scaled):
            s += cp * cq * g / (2.0 * (ap + aq) ** m)
    inv = 1.0 / math.sqrt(s)
    return [c * inv for c in scaled]


def split_general_contractions(shells):
    """PySCF ``_basis`` entries of one species, ``[l, (kappa,) [exp, c_1, ..., c_n], ...]``, as single-column shells
    ``[l, [exp, c], ...]`` in PySCF's AO order: a generally contracted shell (n coefficient columns over one set of exponents — every
    all-electron cc-pVXZ set) is n contracted shells that share their exponents, and its AOs are numbered contraction by contraction
    (libcint: index = contraction * (2l+1) + m), each column normalised on its own (``gto_norm`` + ``_nomalize_contracted_ao``: what
    ``_shell_norm`` does).  Primitives whose coefficient in a column is exactly zero are left out of that column: they contribute nothing
    to its value or its norm.  The reference's default AO path (orbitals.py:46-51, ``mol.eval_gto``) takes any such basis; its in-repo
    evaluator (numba/gto.py:443-455), the oracle and the device take the split shells."""
    out = []
    for sh in shells:
        l, rows = int(sh[0]), list(sh[1:])
        if rows and not hasattr(rows[0], "__len__"):  # [l, kappa, [exp, c], ...]
            if int(rows[0]) != 0:
                raise NotImplementedError("shells with kappa != 0 (spinor basis sets) are not supported")
            rows = rows[1:]
        if not rows:
            raise ValueError("shell without primitives")
        ncol = len(rows[0]) - 1
        if ncol < 1 or any(len(p) != ncol + 1 for p in rows):
            raise ValueError("every primitive of a shell needs one exponent and the same number of coefficients")
        for k in range(ncol):
            prims = [[float(p[0]), float(p[1 + k])] for p in rows if float(p[1 + k]) != 0.0]
            if not prims:
                raise ValueError("contraction column without a non-zero coefficient")
            out.append([l] + prims)
    return out


def basis_tables(mol):
    shell_atom, shell_l, prim_off, ao_off, pexp, pcoef = [], [], [0], [], [], []
    nao = 0
    for ia in range(mol.natm):
        for sh in mol._basis[mol.atom_pure_symbol(ia)]:
            l = int(sh[0])
            if len(sh[1]) != 2:  # (systems.Mol splits general contractions when it is built: split_general_contractions)
                raise ValueError("mol._basis holds a generally contracted shell: pass it through tables.split_general_contractions")
            exps = [float(p[0]) for p in sh[1:]]
            coefs = _shell_norm(l, exps, [float(p[1]) for p in sh[1:]])
            shell_atom.append(ia)
            shell_l.append(l)
            ao_off.append(nao)
            nao += 2 * l + 1
            pexp += exps
            pcoef += coefs
            prim_off.append(len(pexp))
    i32 = lambda v: np.asarray(v, dtype=np.int32)
    return dict(shell_atom=i32(shell_atom), shell_l=i32(shell_l), shell_prim_off=i32(prim_off), shell_ao_off=i32(ao_off),
                prim_exp=np.asarray(pexp, float), prim_coef=np.asarray(pcoef, float), nao=nao)


def pack_determinants(nelec, determinants, tol=-1):
    """-> det_coeff (ndet,), occ_up (ndu,nup), occ_dn (ndd,ndn), det_map (2,ndet)."""
    if determinants is None:
        determinants = [(1.0, [list(range(nelec[0])), list(range(nelec[1]))])]
    coef, occ, dmap = [], [[], []], [[], []]
    for wt, spin_occ in determinants:
        if abs(wt) <= tol:
            continue
        coef.append(float(wt))
        for s in (0, 1):
            o = [int(i) for i in spin_occ[s]]
            if len(o) != nelec[s]:
                raise ValueError("determinant occupation does not match the electron count")
            if o not in occ[s]:
                occ[s].append(o)
            dmap[s].append(occ[s].index(o))
    occ_arr = [np.asarray(occ[s], dtype=np.int32).reshape(len(occ[s]), nelec[s]) for s in (0, 1)]
    return np.asarray(coef, float), occ_arr[0], occ_arr[1], np.asarray(dmap, dtype=np.int32)


def ecp_tables(mol):
    atoms, chan_off, term_off, tn, te, tc = [], [0], [0], [], [], []
    for ia in range(mol.natm):
        sym = mol.atom_pure_symbol(ia)
        if sym not in mol._ecp:
            continue
        chans = {int(l): terms for l, terms in mol._ecp[sym][1]}
        order = sorted(k for k in chans if k >= 0) + [-1]  # non-local l=0.. first, local last
        if order[:-1] != list(range(len(order) - 1)) or -1 not in chans:
            raise NotImplementedError("ECP needs a local channel and contiguous non-local channels l=0..lmax")
        atoms.append(ia)
        for l in order:
            for idx, expand in enumerate(chans[l]):
                for ex, co in expand:
                    tn.append(idx - 2)
                    te.append(float(ex))
                    tc.append(float(co))
            term_off.append(len(tn))
        chan_off.append(len(term_off) - 1)
    i32 = lambda v: np.asarray(v, dtype=np.int32)
    return dict(ecp_atom=i32(atoms), ecp_chan_off=i32(chan_off), ecp_term_off=i32(term_off), ecp_term_n=i32(tn),
                ecp_term_exp=np.asarray(te, float), ecp_term_coef=np.asarray(tc, float))


def jastrow_basis_arrays(basis):
    """basis: list of objects with ``.kind`` (0 PolyPade / 1 CutoffCusp), ``.param`` and ``.rcut``."""
    if not basis:
        return np.zeros(0, np.int32), np.zeros(0), 0.0
    rcut = float(basis[0].rcut)
    for b in basis:
        if float(b.rcut) != rcut:
            raise ValueError("all functions of one Jastrow basis must share rcut (func3d.py:289-291)")
    return np.asarray([b.kind for b in basis], np.int32), np.asarray([b.param for b in basis], float), rcut
