"""Host-side set-up of periodic Slater determinants.

The reference evaluates Bloch orbitals of a *primitive* cell at the k-points that fold onto the supercell's twist:
lattice-summed AOs times Bloch phases per k (``pyqmc/wf/numba/pbcgto.py:99-653``), a wrap phase for the
primitive-cell image the electron sits in (``pyqmc/wf/orbitals.py:192-219``) and one MO block per k (:221-239).

For zero supercell twist and real Bloch phases the same numbers are the *Gamma-point* orbitals of the supercell:
writing a primitive translation as ``L = T_c + Lambda`` (T_c one of the det(S) primitive translations inside the
supercell, Lambda a supercell lattice vector) gives

    psi_{k,n}(r) = sum_{a,mu} sum_c e^{i k.T_c} C_k[a mu, n]  sum_Lambda phi_{a mu}(r - R_a - T_c - Lambda)

i.e. AOs centred on the supercell's atoms ``R_a + T_c`` (the very atom list ``supercell.get_supercell`` builds,
``pyqmc/pbc/supercell.py:45-78``), lattice-summed over the supercell lattice, with the k-phase folded into a real
coefficient matrix.  That is what the device evaluates (one phase-free lattice sum per point, accumulators small
enough for registers); this module builds those tables.  Cut-offs are the reference's (``max_Ls``,
pbcgto.py:549-591); the image list per atom is every supercell translation that can come within the atom's cut-off
of a point in the cell — a superset of what the cut-off lets through, as the reference's ``num_Ls`` is meant to be.
"""

import numpy as np

from . import tables
from .systems import Cell


# ------------------------------------------------------------------ supercells (pyqmc/pbc/supercell.py)
def _points_in_unit_box(S, transpose):
    S = np.asarray(S, dtype=float)
    box = np.stack([x.ravel() for x in np.meshgrid(*[[0, 1]] * 3, indexing="ij")]).T @ (S.T if transpose else S)
    rng = [np.arange(lo, hi) for lo, hi in zip(box.min(axis=0), box.max(axis=0))]
    mesh = np.stack([x.ravel() for x in np.meshgrid(*rng, indexing="ij")]).T
    frac = mesh @ (np.linalg.inv(S).T if transpose else np.linalg.inv(S))
    return frac[np.all((frac >= 0) & (frac < 1 - 1e-12), axis=1)]


def get_supercell_copies(latvec, S):
    """Primitive translations inside the supercell (supercell.py:32-42)."""
    return np.linalg.multi_dot((_points_in_unit_box(S, False), np.asarray(S, dtype=float), latvec))


def get_supercell_kpts(supercell):
    """k-points of the primitive cell that are Gamma of the supercell (supercell.py:18-29)."""
    rec = np.linalg.inv(supercell.original_cell.lattice_vectors()).T * 2 * np.pi
    return _points_in_unit_box(supercell.S, True) @ rec


def get_supercell(cell, S):
    """Supercell with lattice ``S @ cell.a``; atoms ordered primitive-atom-major like supercell.py:57-60."""
    S = np.asarray(S, dtype=float)
    copies = get_supercell_copies(cell.lattice_vectors(), S)
    names, xyz = [], []
    for name, r in zip(cell._names, cell.atom_coords()):
        for R in copies:
            names.append(name)
            xyz.append(r + R)
    scale = abs(int(round(np.linalg.det(S))))
    ne = (cell.nelec[0] * scale, cell.nelec[1] * scale)
    sup = Cell(names, xyz, S @ cell.lattice_vectors(), nelec=ne, basis=cell._basis, ecp=cell._ecp,
               charges=np.repeat(cell.atom_charges(), scale))
    sup.original_cell, sup.S, sup.scale = cell, S, scale
    return sup


# ------------------------------------------------------------------ lattice sums
def lattice_points_within(latvec, radius):
    """All lattice translations n @ latvec with norm <= radius, sorted by norm (what the reference takes from
    ``cell.get_lattice_Ls`` and sorts, pbcgto.py:602-603)."""
    heights = 1.0 / np.linalg.norm(np.linalg.inv(latvec), axis=0)  # lattice-plane spacings
    nmax = np.ceil(radius / heights).astype(int) + 1
    grid = np.stack([g.ravel() for g in np.meshgrid(*[np.arange(-n, n + 1) for n in nmax], indexing="ij")]).T
    L = grid @ latvec
    norm = np.linalg.norm(L, axis=1)
    keep = norm <= radius
    order = np.argsort(norm[keep], kind="stable")
    return L[keep][order]


def cell_diameter(latvec):
    """Largest distance between two points of the parallelepiped (its longest body diagonal)."""
    combos = np.array([[1.0, 1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 1.0, -1.0]])
    return float(np.max(np.linalg.norm(combos @ latvec, axis=1)))


def gto_cutoffs(bt, expcutoff):
    """r^2 cut-offs per atom and per shell (pbcgto.py:565-583) from ``tables.basis_tables`` output."""
    nshell = len(bt["shell_l"])
    lcut, acut = np.zeros(nshell), np.zeros(int(bt["shell_atom"].max()) + 1)
    for i in range(nshell):
        p0, p1 = bt["shell_prim_off"][i], bt["shell_prim_off"][i + 1]
        exps, log_c, l = bt["prim_exp"][p0:p1], np.log(np.abs(bt["prim_coef"][p0:p1])), int(bt["shell_l"][i])
        lconst = 0.0 if l == 0 else 0.5 * np.log(0.5 * l / np.amin(exps)) * l
        lcut[i] = np.amax((expcutoff + log_c + lconst) / exps)
        acut[bt["shell_atom"][i]] = max(acut[bt["shell_atom"][i]], lcut[i])
    return acut, lcut


def reference_num_Ls(bt, Ls, lvecs, expcutoff):
    """How many of the norm-sorted translations ``Ls`` the reference looks at per atom (``max_Ls``,
    pbcgto.py:549-591: the last translation whose Gaussian tail, seen from the cell's far corner, is above the
    cut-off).  NOTE this is not a superset of what the r^2 cut-offs let through elsewhere in the cell (dropped terms
    reach ~1e-4 at the default precision, tests/golden/make_golden.py:g_pbc_slater prints it); to reproduce the
    reference's numbers the device therefore applies the same membership rule (``member`` below)."""
    combos = np.array([[1.0, 1.0, 1.0], [-1.0, 1.0, 1.0], [1.0, -1.0, 1.0], [1.0, 1.0, -1.0]])
    vecs = combos @ lvecs
    v = vecs[np.argmax(np.sum(vecs**2, axis=-1))] / 2  # max_distance_in_cell, pbcgto.py:509-520
    r2 = np.sum((v - Ls) ** 2, axis=-1)
    num = np.zeros(int(bt["shell_atom"].max()) + 1, dtype=np.int64)
    for i in range(len(bt["shell_l"])):
        p0, p1 = bt["shell_prim_off"][i], bt["shell_prim_off"][i + 1]
        exps, log_c, l = bt["prim_exp"][p0:p1], np.log(np.abs(bt["prim_coef"][p0:p1])), int(bt["shell_l"][i])
        with np.errstate(divide="ignore"):
            min_exp = np.amin(exps[None, :] * r2[:, None] - log_c[None, :] - 0.5 * np.log(r2)[:, None] * l, axis=1)
        where = np.where(min_exp < expcutoff)[0]
        ia = bt["shell_atom"][i]
        num[ia] = max(num[ia], where.max() + 1 if len(where) else 1)
    return num


def periodic_tables(supercell, eval_gto_precision=None, Ls_prim=None, image_rule="reference"):
    """Lattice-sum tables for the supercell's Gamma-point AOs.

    Every AO centred on supercell atom A = (primitive atom a, copy c) is summed over the supercell translations
    ``Ls[j]``, j < ``num_Ls[A]`` — all that can come within the atom's cut-off of the (folded) displacement
    point - atom — subject to
    the reference's r^2 cut-offs (``atom_cut``, ``shell_cut``; pbcgto.py:565-583, expcutoff :604).  With
    ``image_rule="reference"`` an image is additionally kept only if the reference would have looked at it: the
    reference folds the point into the primitive cell (wrap W, orbitals.py:201) and sums the first ``num_Ls[a]``
    entries of its norm-sorted primitive translation list (pbcgto.py:603-616), so the primitive translation
    ``n(T_c) + n(Lambda) - W`` must be one of those.  ``member`` is that set as a byte grid over integer triplets
    per class of atoms, ``atom_n`` / ``img_n`` are n(T_c) / n(Lambda).  ``Ls_prim``: the reference's sorted list
    (from ``cell.get_lattice_Ls`` there); by default every primitive translation up to 30 bohr + cell diameter.
    ``image_rule="complete"`` keeps every image inside the cut-offs (translation invariant, more accurate)."""
    precision = 1e-2 if eval_gto_precision is None else eval_gto_precision  # orbitals.py:150, pbcgto.py:599
    expcutoff = -3.5 * np.log(precision)  # pbcgto.py:604
    prim = supercell.original_cell
    S = np.asarray(supercell.S, dtype=float)
    bt = tables.basis_tables(supercell)
    acut, lcut = gto_cutoffs(bt, expcutoff)
    lat, lprim = supercell.lattice_vectors(), prim.lattice_vectors()
    # the device folds (point - atom) into the cell-centred parallelepiped first, so an image only matters if it is
    # within sqrt(cut) + half the longest body diagonal of the origin
    reach = (np.sqrt(acut) + 0.5 * cell_diameter(lat)) * (1 + 1e-9)
    Ls = lattice_points_within(lat, reach.max())
    norms = np.linalg.norm(Ls, axis=1)
    num = np.array([int(np.searchsorted(norms, r, side="right")) for r in reach], dtype=np.int32)
    img_n = np.rint(Ls @ np.linalg.inv(lprim)).astype(np.int32)
    ncopy = supercell.scale
    copies = get_supercell_copies(lprim, S)
    atom_n = np.tile(np.rint(copies @ np.linalg.inv(lprim)).astype(np.int32), (prim.natm, 1))
    out = {"Ls": Ls, "num_Ls": num, "atom_cut": acut, "shell_cut": lcut, "img_n": img_n, "atom_n": atom_n,
           "lattice_prim": lprim, "member": None, "member_class": np.zeros(supercell.natm, dtype=np.int32), "member_M": 0}
    if image_rule == "complete":
        return out
    if image_rule != "reference":
        raise ValueError("image_rule must be 'reference' or 'complete'")
    if Ls_prim is None:
        Ls_prim = lattice_points_within(lprim, 30.0 + cell_diameter(lprim))
    bt_prim = tables.basis_tables(prim)
    num_prim = reference_num_Ls(bt_prim, Ls_prim, lprim, expcutoff)
    n_prim = np.rint(Ls_prim @ np.linalg.inv(lprim)).astype(int)
    M = int(np.abs(n_prim[: num_prim.max()]).max())
    classes = sorted(set(int(n) for n in num_prim))
    member = np.zeros((len(classes), 2 * M + 1, 2 * M + 1, 2 * M + 1), dtype=np.uint8)
    for ci, n in enumerate(classes):
        t = n_prim[:n] + M
        member[ci, t[:, 0], t[:, 1], t[:, 2]] = 1
    out["member"], out["member_M"] = member, M
    out["member_class"] = np.repeat([classes.index(int(n)) for n in num_prim], ncopy).astype(np.int32)
    out["num_Ls_prim"] = num_prim
    return out


# ------------------------------------------------------------------ folding k-point MOs onto the supercell
def fold_mo_coeff(supercell, kpts, mo_coeff):
    """mo_coeff[s][k] (nao_prim, nmo_k) at primitive k-points -> (2)[nao_super, sum_k nmo_k] matrices (real when every Bloch
    phase and coefficient is, complex otherwise) in the AO order of ``tables.basis_tables(supercell)`` (atoms primitive-atom-major, copies inside)."""
    prim = supercell.original_cell
    kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
    copies = get_supercell_copies(prim.lattice_vectors(), supercell.S)
    if len(kpts) != supercell.scale:
        raise ValueError(f"found {len(kpts)} k-points but the supercell folds {supercell.scale} (pyscftools.py:163-166)")
    phase = np.exp(1j * copies @ kpts.T)  # (ncopy, nk); a common twist stays in the AOs' lattice sums (see common_twist)
    cplx = np.abs(phase.imag).max() > 1e-9 or any(np.iscomplexobj(m) and np.abs(np.imag(m)).max(initial=0.0) > 1e-12
                                                   for s in (0, 1) for m in mo_coeff[s])
    if not cplx:
        phase = phase.real
    nao_atom = [sum(2 * sh[0] + 1 for sh in prim._basis[n]) for n in prim._names]
    off = np.concatenate([[0], np.cumsum(nao_atom)])
    out = []
    for s in (0, 1):
        blocks = []
        for k in range(len(kpts)):
            C = np.asarray(mo_coeff[s][k]) if cplx else np.real(np.asarray(mo_coeff[s][k]))
            rows = [phase[c, k] * C[off[a] : off[a + 1]] for a in range(len(nao_atom)) for c in range(len(copies))]
            blocks.append(np.concatenate(rows, axis=0))
        out.append(np.concatenate(blocks, axis=1))
    return out


def unfold_mo_gradient(supercell, kpts, d_super, nmo_k):
    """Chain rule of ``fold_mo_coeff``: derivatives w.r.t. the folded supercell coefficients
    ``d_super`` (..., nao_super, sum_k nmo_k) -> derivatives w.r.t. the per-k blocks, concatenated over k like the
    reference's ``mo_coeff_alpha`` parameter (..., nao_prim, sum_k nmo_k) (orbitals.py:154-160, slater.py:511-527):
    d/dC_k[(a,m), n] = sum_c e^{i k . T_c} d/dC_super[(a,c,m), (k,n)] (C_super = e^{i k . T_c} C_k is holomorphic in C_k;
    real phases and derivatives stay real)."""
    prim = supercell.original_cell
    kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
    copies = get_supercell_copies(prim.lattice_vectors(), supercell.S)
    phase = np.exp(1j * copies @ kpts.T)
    if np.abs(phase.imag).max() <= 1e-9 and not np.iscomplexobj(d_super):
        phase = phase.real
    nao_atom = [sum(2 * sh[0] + 1 for sh in prim._basis[n]) for n in prim._names]
    ncopy = len(copies)
    lead = d_super.shape[:-2]
    out = np.zeros(lead + (sum(nao_atom), d_super.shape[-1]), dtype=np.result_type(phase.dtype, d_super.dtype))
    col = np.concatenate([[0], np.cumsum(nmo_k)])
    row = prow = 0
    for na in nao_atom:
        blk = d_super[..., row : row + ncopy * na, :].reshape(lead + (ncopy, na, d_super.shape[-1]))
        for k in range(len(kpts)):
            out[..., prow : prow + na, col[k] : col[k + 1]] = np.einsum("c,...cmn->...mn", phase[:, k], blk[..., col[k] : col[k + 1]])
        row += ncopy * na
        prow += na
    return out


def common_twist(supercell, kpts):
    """The twist the k-points share: k modulo the supercell's reciprocal lattice, as a cartesian vector with fractional
    components in [-1/2, 1/2) — or None when it is zero.  Raises if the k-points do not share one."""
    lat = supercell.lattice_vectors()
    frac = np.asarray(kpts, dtype=float).reshape(-1, 3) @ lat.T / (2 * np.pi)
    red = frac - np.floor(frac + 0.5 + 1e-12)
    if np.abs(red - red[0]).max() > 1e-9:
        raise ValueError("the k-points do not share one supercell twist")
    if np.abs(red[0]).max() < 1e-9:
        return None
    return red[0] @ (2 * np.pi * np.linalg.inv(lat).T)


class KMeanField:
    """Duck-typed k-point mean field: ``kpts`` (nk,3), ``mo_coeff[s][k]`` (nao_prim, nmo), ``mo_occ[s][k]`` (nmo,)."""

    def __init__(self, kpts, mo_coeff, mo_occ):
        self.kpts = np.asarray(kpts, dtype=float).reshape(-1, 3)
        self.mo_coeff, self.mo_occ = mo_coeff, mo_occ

    def to_uhf(self, *a):
        return self


def random_kmf(supercell, seed=20260928, nvirt=0, complex_coeff=False, twist=None):
    """Seeded Bloch coefficients at the supercell's Gamma-compatible k-points, electrons spread evenly over the
    k-points (what an insulator's KRHF gives).  ``complex_coeff``: randn + i randn before orthonormalisation."""
    prim = supercell.original_cell
    kpts = get_supercell_kpts(supercell)
    if twist is not None:  # fractional twist in units of the supercell's reciprocal lattice vectors
        kpts = kpts + np.asarray(twist, dtype=float) @ supercell.reciprocal_vectors()
    rng = np.random.default_rng(seed)
    nao = prim.nao()
    mo, occ = [[], []], [[], []]
    for s in (0, 1):
        per_k, rem = divmod(supercell.nelec[s], len(kpts))
        for k in range(len(kpts)):
            n = per_k + (k < rem)
            a = rng.standard_normal((nao, nao))
            if complex_coeff:
                a = a + 1j * rng.standard_normal((nao, nao))
            q, _ = np.linalg.qr(a)
            mo[s].append(q[:, : n + nvirt])
            o = np.zeros(n + nvirt)
            o[:n] = 1.0
            occ[s].append(o)
    return KMeanField(kpts, mo, occ)
