"""Oracle: Slater / JastrowSpin / MultiplyWF wave-function protocol on the CPU
(test infrastructure; NumPy restatement of the reference algorithm).

Protocol (``doc/source/wavefunction.rst:5-40``): ``recompute``, ``value``,
``gradient``, ``gradient_value``, ``gradient_laplacian``, ``testvalue``,
``updateinternals``.  ``configs`` / ``epos`` arguments are duck-typed objects with a
``.configs`` array, exactly as in the reference.

Reference lines followed are cited per method.  Structure differs (flat
tables instead of PySCF objects, explicit loops over unique determinants),
arithmetic does not.
"""

import numpy as np

from . import gto, jastrow_basis


def _phase(x):
    return np.sign(x) if not np.iscomplexobj(x) else x / np.abs(x)


class _ManyMixin:
    def testvalue_many(self, es, epos, mask=None):
        """``testvalue_many`` of every factor (slater.py:448-460, jastrowspin.py:421-455, three_body_jastrow.py:343-372,
        multiplywf.py:112-114): column i is ``testvalue(es[i], epos)`` — one auxiliary position, several electrons."""
        return np.stack([np.asarray(self.testvalue(int(e), epos, mask)[0]) for e in np.atleast_1d(es)], axis=1)


class Slater(_ManyMixin):
    """Multi-determinant Slater wave function (``pyqmc/wf/slater.py:97-460``).

    mol: Mol-like; mo_coeff: (2, nao, nmo_s) per spin (already truncated to the used
    columns, pyscftools.py:181-183); determinants: [(coef,[occ_up,occ_dn]),...] or None
    for the aufbau determinant (pyscftools.py:206-218); packing per
    determinant_tools.py:39-71."""

    def __init__(self, mol, mo_coeff, determinants=None):
        self._nelec = tuple(mol.nelec)
        self.table = gto.AOTable(mol)
        if determinants is None:
            determinants = [(1.0, [list(range(self._nelec[0])), list(range(self._nelec[1]))])]
        coefs, occup, dmap = [], [[], []], [[], []]
        for wt, occ in determinants:
            coefs.append(wt)
            for s in (0, 1):
                o = [int(i) for i in occ[s]]
                if o not in occup[s]:
                    occup[s].append(o)
                dmap[s].append(occup[s].index(o))
        self._det_occup = [np.asarray(occup[s], dtype=int).reshape(len(occup[s]), self._nelec[s]) for s in (0, 1)]
        self._det_map = np.asarray(dmap, dtype=int)
        nmo = [int(self._det_occup[s].max(initial=-1)) + 1 for s in (0, 1)]
        self.parameters = {
            "det_coeff": np.asarray(coefs, dtype=float),
            "mo_coeff_alpha": np.array(mo_coeff[0][:, : nmo[0]]),
            "mo_coeff_beta": np.array(mo_coeff[1][:, : nmo[1]]),
        }
        self.dtype = float

    @classmethod
    def periodic(cls, supercell, kpts, mo_coeff, Ls, determinants=None, precision=1e-2):
        """Slater determinant of Bloch orbitals (slater.py:181-225 with PBCOrbitalEvaluatorKpoints).  mo_coeff[s][k]
        (nao_prim, nmo_k); determinants index the k-concatenated MO list (determinant_tools.flatten_determinants)."""
        from .pbc import PeriodicOrbitals

        orb = PeriodicOrbitals(supercell, kpts, mo_coeff, Ls, precision)
        self = cls(supercell, [np.concatenate(orb.mo[s], axis=1) for s in (0, 1)], determinants)
        self.dtype = complex if orb.complex else float  # slater.py:212-216
        self._orb = orb  # the MO blocks live in orb.mo; parameters["mo_coeff_*"] are their concatenation (orbitals.py:157-160)
        return self

    # -- helpers ---------------------------------------------------------
    def _spin(self, e):
        s = int(e >= self._nelec[0])
        return s, e - s * self._nelec[0]

    def _r(self, obj):
        """Positions for the orbital evaluator: periodic orbitals take the TRUE (unfolded) coordinates, from which the
        evaluator recovers the wrap counters the reference's ``aos`` reads off ``configs.wrap`` (orbitals.py:203-209)."""
        x = np.asarray(obj.configs, dtype=float)
        if getattr(self, "_orb", None) is not None and hasattr(obj, "wrap"):
            x = x + np.asarray(obj.wrap, dtype=float) @ obj.lvecs
        return x

    def _mo(self, pts, s, ncomp):
        if getattr(self, "_orb", None) is not None:
            ao = self._orb.aos(pts, ncomp)
            return ao, self._orb.mos(ao, s)
        c = self.parameters["mo_coeff_alpha" if s == 0 else "mo_coeff_beta"]
        ao = gto.eval_ao(self.table, pts, ncomp)
        return ao, gto.eval_mo(ao, c)

    def _det_weights(self, mask=None):
        """c_D * sign_up*sign_dn*exp(logup+logdn-ref) per full determinant D
        (slater.py:313-325; reference uses the global max as ref — any ref cancels)."""
        up, dn = self._dets[0], self._dets[1]
        if mask is not None:
            up, dn = up[:, mask], dn[:, mask]
        lu = up[1][:, self._det_map[0]]
        ld = dn[1][:, self._det_map[1]]
        ref = np.amax(self._dets[0][1]) + np.amax(self._dets[1][1])
        arr = up[0][:, self._det_map[0]] * dn[0][:, self._det_map[1]] * np.exp(lu + ld - ref)
        return arr * self.parameters["det_coeff"][None, :]  # (W, D)

    # -- protocol --------------------------------------------------------
    def recompute(self, configs):
        """slater.py:227-260: AO -> MO -> per unique determinant slogdet and inverse."""
        x = self._r(configs)
        nconf, nelec, _ = x.shape
        self._x_last = x.copy()
        self._dets, self._inverse = [], []
        for s in (0, 1):
            b, e = self._nelec[0] * s, self._nelec[0] + self._nelec[1] * s
            _, mo = self._mo(x[:, b:e].reshape(-1, 3), s, 1)
            mo = mo[0].reshape(nconf, e - b, mo.shape[-1])
            mats = np.stack([mo[:, :, occ] for occ in self._det_occup[s]], axis=1)  # (W,D,n,n) [elec,orb]
            sign, logdet = np.linalg.slogdet(mats)
            self._dets.append(np.array([sign, logdet]))
            inv = np.zeros_like(mats)
            ok = np.isfinite(logdet)
            inv[ok] = np.linalg.inv(mats[ok])
            self._inverse.append(inv)  # [orbital j, electron i]
        return self.value()

    def value(self):
        """determinant_tools.py:74-88."""
        ref = np.amax(self._dets[0][1]) + np.amax(self._dets[1][1])
        tot = self._det_weights().sum(axis=1)
        with np.errstate(divide="ignore", invalid="ignore"):
            sign = np.nan_to_num(tot / np.abs(tot))
            logv = np.nan_to_num(np.log(np.abs(tot)) + ref)
        return sign, logv

    def pgradient(self):
        """slater.py:462-542: d Psi/Psi w.r.t. det_coeff (W, ndet) and mo_coeff_* (W, nao, nmo) — needs the walker
        coordinates of the last recompute (the reference keeps the AO values ``_aovals`` for this)."""
        sign, logv = self.value()
        lu = self._dets[0][1][:, self._det_map[0]]
        ld = self._dets[1][1][:, self._det_map[1]]
        with np.errstate(divide="ignore", invalid="ignore"):
            ddet = np.where(sign[:, None] != 0,
                            self._dets[0][0][:, self._det_map[0]] * self._dets[1][0][:, self._det_map[1]]
                            * np.exp(lu + ld - logv[:, None]) / sign[:, None], 0.0)
        out = {"det_coeff": ddet}
        W = len(sign)
        for s, name in ((0, "mo_coeff_alpha"), (1, "mo_coeff_beta")):
            b, e = self._nelec[0] * s, self._nelec[0] + self._nelec[1] * s
            ao, _ = self._mo(self._x_last[:, b:e].reshape(-1, 3), s, 1)
            if getattr(self, "_orb", None) is not None:  # periodic: per-k AO blocks (nk, 1, P, nao_prim) and per-k columns (orbitals.py:239-254)
                aos = [ao[k][0].reshape(W, e - b, -1) for k in range(len(self._orb.kpts))]
                split = np.cumsum([0] + [m.shape[1] for m in self._orb.mo[s]])
            else:
                aos = [ao[0].reshape(W, e - b, -1)]  # (W, n, nao)
                split = np.array([0, self.parameters[name].shape[1]])
            nmo = int(split[-1])
            g = np.zeros((W, aos[0].shape[-1], nmo), dtype=np.result_type(ddet.dtype, aos[0].dtype, self._inverse[s].dtype))
            for di, coeff in enumerate(self.parameters["det_coeff"]):
                u = self._det_map[s][di]
                for col, m in enumerate(self._det_occup[s][u]):  # _testcol: sum_e ao[w,e,a] inverse[w,u,col,e]
                    k = int(np.searchsorted(split, m, side="right") - 1)
                    g[:, :, m] += coeff * ddet[:, di, None] * np.einsum("wea,we->wa", aos[k], self._inverse[s][:, u, col, :])
            if g.size:
                out[name] = g
        return out

    def _row_ratios(self, e, mo_rows, mask=None):
        """Ratio of replacing row ``e`` by mo_rows[...] for each leading component.

        mo_rows: (..., Wm, nmo) with arbitrary leading axes; returns (..., Wm).
        slater.py:301-380 (_testrow/_testrowderiv)."""
        s, eeff = self._spin(e)
        inv = self._inverse[s] if mask is None else self._inverse[s][mask]
        col = inv[..., eeff]  # (Wm, D_s, n) over orbital j
        rows = mo_rows[..., self._det_occup[s]]  # (..., Wm, D_s, n)
        rat = np.einsum("...wdj,wdj->...wd", rows, col)
        wts = self._det_weights(mask)  # (Wm, D)
        numer = np.einsum("...wd,wd->...w", rat[..., self._det_map[s]], wts)
        return numer / wts.sum(axis=1)

    def gradient_value(self, e, epos):
        """slater.py:403-418."""
        s, _ = self._spin(e)
        ao, mo = self._mo(self._r(epos), s, 4)
        rat = self._row_ratios(e, mo)
        with np.errstate(divide="ignore", invalid="ignore"):
            grad = rat[1:] / rat[0]
        grad[~np.isfinite(grad)] = 0.0
        val = rat[0].copy()
        val[~np.isfinite(val)] = 1.0
        return grad, val, (ao[0], mo[0])

    def gradient(self, e, epos):
        """slater.py:390-401."""
        s, _ = self._spin(e)
        _, mo = self._mo(self._r(epos), s, 4)
        rat = self._row_ratios(e, mo)
        return rat[1:] / rat[0]

    def gradient_laplacian(self, e, epos):
        """slater.py:420-427."""
        s, _ = self._spin(e)
        _, mo = self._mo(self._r(epos), s, 5)
        rat = self._row_ratios(e, mo)
        rat = rat / rat[:1]
        return rat[1:4], rat[4]

    def testvalue(self, e, epos, mask=None):
        """slater.py:429-446; epos (W,3) or (W,naip,3); mask selects walkers."""
        s, _ = self._spin(e)
        x = self._r(epos) if mask is None else self._r(epos)[mask]
        ao, mo = self._mo(x.reshape(-1, 3), s, 1)
        mo0 = mo[0].reshape(x.shape[:-1] + (-1,))
        if x.ndim == 3:
            rat = self._row_ratios(e, np.moveaxis(mo0, 1, 0), mask)  # (naip, Wm)
            return rat.T, (ao[0], mo[0])
        return self._row_ratios(e, mo0, mask), (ao[0], mo[0])

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        """slater.py:262-291 with Sherman-Morrison slater.py:88-94."""
        s, eeff = self._spin(e)
        nconf = epos.configs.shape[0]
        mask = np.ones(nconf, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        if np.any(np.isinf(self._dets[s][1])):
            self.recompute(configs)
            return
        if saved_values is None:
            _, mo = self._mo(self._r(epos)[mask], s, 1)
            mo = mo[0]
        else:
            mo = saved_values[1][mask]
        vec = mo[:, self._det_occup[s]]  # (Wm, D, n)
        inv = self._inverse[s][mask]
        tmp = np.einsum("wdk,wdkj->wdj", vec, inv)
        ratio = tmp[:, :, eeff]
        invr = inv[:, :, :, eeff] / ratio[:, :, None]
        inv = inv - np.einsum("wdi,wdj->wdij", invr, tmp)
        inv[:, :, :, eeff] = invr
        self._inverse[s][mask] = inv
        self._dets[s][0][mask] *= _phase(ratio)
        self._dets[s][1][mask] += np.log(np.abs(ratio))
        self._x_last[mask, e] = self._r(epos)[mask]


class JastrowSpin(_ManyMixin):
    """One- and two-body spin Jastrow e^U (``pyqmc/wf/jastrowspin.py:20-419``).

    a_basis / b_basis: lists of ("pade",beta)/("cusp",gamma) with common rcut."""

    def __init__(self, mol, a_basis, b_basis, rcut):
        self._nup, self._ndn = mol.nelec
        self._nelec = self._nup + self._ndn
        self.atoms = np.asarray(mol.atom_coords(), dtype=float)
        self.a_basis, self.b_basis, self.rcut = list(a_basis), list(b_basis), float(rcut)
        # periodic cell: every displacement goes through the minimal-image rule (configs.dist, jastrowspin.py:82-98)
        if hasattr(mol, "a"):
            from .pbc import minimal_image

            self._mi = minimal_image(mol.lattice_vectors())
        else:
            self._mi = lambda d: d
        self.parameters = {
            "bcoeff": np.zeros((len(b_basis), 3)),
            "acoeff": np.zeros((len(self.atoms), len(a_basis), 2)),
        }
        self.dtype = float

    def _a(self, d, want):
        d = self._mi(d)
        return jastrow_basis.evaluate(self.a_basis, self.rcut, d, np.linalg.norm(d, axis=-1), want)

    def _b(self, d, want):
        d = self._mi(d)
        return jastrow_basis.evaluate(self.b_basis, self.rcut, d, np.linalg.norm(d, axis=-1), want)

    def _others(self, e):
        return np.arange(self._nelec) != e

    def _sep(self, e):
        return self._nup - int(e < self._nup)

    def recompute(self, configs):
        """jastrowspin.py:56-109."""
        x = configs.configs
        self._x = x.copy()
        W, N = x.shape[:2]
        nup = self._nup
        self._a_partial = np.zeros((N, W, len(self.atoms), len(self.a_basis)))
        self._b_partial = np.zeros((N, W, len(self.b_basis), 2))
        for e in range(N):
            self._a_partial[e] = self._a(x[:, e, None, :] - self.atoms[None], "value")
            bv = self._b(x[:, e, None, :] - x[:, self._others(e)], "value")
            sep = self._sep(e)
            self._b_partial[e, :, :, 0] = bv[:, :sep].sum(axis=1)
            self._b_partial[e, :, :, 1] = bv[:, sep:].sum(axis=1)
        self._avalues = np.stack([self._a_partial[:nup].sum(axis=0), self._a_partial[nup:].sum(axis=0)], axis=-1)
        self._bvalues = np.zeros((W, len(self.b_basis), 3))

        def pairs(xa, xb, same):
            if same:
                iu, ju = np.triu_indices(xa.shape[1], k=1)
                return xa[:, iu] - xa[:, ju]
            return (xb[:, None, :, :] - xa[:, :, None, :]).reshape(W, -1, 3)

        for j, d in enumerate([pairs(x[:, :nup], None, True), pairs(x[:, :nup], x[:, nup:], False), pairs(x[:, nup:], None, True)]):
            self._bvalues[:, :, j] = self._b(d, "value").sum(axis=1)
        return self.value()

    def value(self):
        """jastrowspin.py:251-255."""
        u = np.sum(self._bvalues * self.parameters["bcoeff"], axis=(2, 1))
        u += np.einsum("ijkl,jkl->i", self._avalues, self.parameters["acoeff"])
        return np.ones(len(u)), u

    def pgradient(self):
        """jastrowspin.py:457-464: the stored sums."""
        return {"bcoeff": self._bvalues.copy(), "acoeff": self._avalues.copy()}

    def _new_partials(self, e, x_new, mask):
        """a- and b- partial sums of electron e placed at x_new (Wm[,naip],3)
        (jastrowspin.py:139-191)."""
        others = self._x[mask][:, self._others(e)]  # (Wm, N-1, 3)
        if x_new.ndim == 2:
            da = x_new[:, None, :] - self.atoms[None]
            db = x_new[:, None, :] - others
        else:  # aux points: leading axis naip like the reference's moveaxis(d,2,0)
            da = np.moveaxis(x_new, 1, 0)[:, :, None, :] - self.atoms[None, None]
            db = np.moveaxis(x_new, 1, 0)[:, :, None, :] - others[None]
        a_new = self._a(da, "value")
        bv = self._b(db, "value")
        sep = self._sep(e)
        b_new = np.stack([bv[..., :sep, :].sum(axis=-2), bv[..., sep:, :].sum(axis=-2)], axis=-1)
        return a_new, b_new, bv

    def testvalue(self, e, epos, mask=None):
        """jastrowspin.py:387-419."""
        W = epos.configs.shape[0]
        mask = np.ones(W, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        edown = int(e >= self._nup)
        a_new, b_new, bv = self._new_partials(e, epos.configs[mask], mask)
        a_val = np.einsum("...jk,jk->...", a_new - self._a_partial[e][mask], self.parameters["acoeff"][..., edown])
        b_val = np.einsum("...jk,jk->...", b_new - self._b_partial[e][mask], self.parameters["bcoeff"][:, edown : edown + 2])
        val = np.exp(b_val + a_val)
        return (val.T if val.ndim == 2 else val), (a_new, b_new, bv)

    def _grad_terms(self, e, x_new, want):
        nup = self._nup
        eup, edown = int(e < nup), int(e >= nup)
        db = x_new[:, None, :] - self._x[:, self._others(e)]
        da = x_new[:, None, :] - self.atoms[None]
        bg, bs = self._b(db, want)
        ag, as_ = self._a(da, want)
        bc = self.parameters["bcoeff"]
        ac = self.parameters["acoeff"][:, :, edown]
        sep = nup - eup
        grad = np.einsum("b,cbx->xc", bc[:, edown], bg[:, :sep].sum(axis=1))
        grad += np.einsum("b,cbx->xc", bc[:, 1 + edown], bg[:, sep:].sum(axis=1))
        grad += np.einsum("ab,cabx->xc", ac, ag)
        return grad, bs, as_, sep, bc, ac, edown

    def gradient(self, e, epos):
        """jastrowspin.py:257-294."""
        return self._grad_terms(e, epos.configs, "gradient_value")[0]

    def gradient_value(self, e, epos):
        """jastrowspin.py:296-340."""
        grad, bval, aval, sep, bc, ac, edown = self._grad_terms(e, epos.configs, "gradient_value")
        b_new = np.stack([bval[:, :sep].sum(axis=1), bval[:, sep:].sum(axis=1)], axis=-1)
        a_val = np.einsum("...ab,ab->...", aval - self._a_partial[e], ac)
        b_val = np.einsum("...jk,jk->...", b_new - self._b_partial[e], bc[:, edown : edown + 2])
        return grad, np.exp(b_val + a_val), (aval, b_new, bval)

    def gradient_laplacian(self, e, epos):
        """jastrowspin.py:342-385: returns grad U and lap U + |grad U|^2."""
        grad, blap, alap, sep, bc, ac, edown = self._grad_terms(e, epos.configs, "gradient_laplacian")
        lap = np.einsum("ab,cab->c", ac, alap)
        lap += np.einsum("b,cb->c", bc[:, edown], blap[:, :sep].sum(axis=1))
        lap += np.einsum("b,cb->c", bc[:, 1 + edown], blap[:, sep:].sum(axis=1))
        return grad, lap + np.sum(grad**2, axis=0)

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        """jastrowspin.py:111-137 and :221-249."""
        W = self._x.shape[0]
        mask = np.ones(W, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        edown = int(e >= self._nup)
        if saved_values is None:
            a_new, b_new, bv = self._new_partials(e, epos.configs[mask], mask)
        else:
            a_new, b_new, bv = [s[mask] for s in saved_values]
        self._avalues[mask, :, :, edown] += a_new - self._a_partial[e][mask]
        self._bvalues[mask, :, edown : edown + 2] += b_new - self._b_partial[e][mask]
        self._a_partial[e][mask] = a_new
        others = self._others(e)
        old = self._b(self._x[mask, e, None, :] - self._x[mask][:, others], "value")
        diff = bv - old  # (Wm, N-1, nb)
        idx = np.nonzero(others)[0]
        midx = np.nonzero(mask)[0]
        self._b_partial[idx[:, None], midx[None, :], :, edown] += np.moveaxis(diff, 1, 0)
        self._b_partial[e][mask] = b_new
        self._x[mask, e, :] = epos.configs[mask]


class MultiplyWF(_ManyMixin):
    def pgradient(self):
        """multiplywf.py:131-132, as {"wf{i}{key}": array}."""
        return {f"wf{i + 1}{k}": v for i, w in enumerate(self.wf_factors) for k, v in w.pgradient().items()}

    """Product of factors (``pyqmc/wf/multiplywf.py:71-132``)."""

    def __init__(self, *wf_factors):
        self.wf_factors = list(wf_factors)
        self.dtype = complex if any(w.dtype == complex for w in wf_factors) else float
        self.parameters = {f"wf{i + 1}{k}": v for i, w in enumerate(wf_factors) for k, v in w.parameters.items()}

    def recompute(self, configs):
        res = [w.recompute(configs) for w in self.wf_factors]
        return np.prod([r[0] for r in res], axis=0), np.sum([r[1] for r in res], axis=0)

    def value(self):
        res = [w.value() for w in self.wf_factors]
        return np.prod([r[0] for r in res], axis=0), np.sum([r[1] for r in res], axis=0)

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        saved_values = [None] * len(self.wf_factors) if saved_values is None else saved_values
        for w, sv in zip(self.wf_factors, saved_values):
            w.updateinternals(e, epos, configs, mask=mask, saved_values=sv)

    def gradient(self, e, epos):
        return np.sum([w.gradient(e, epos) for w in self.wf_factors], axis=0)

    def testvalue(self, e, epos, mask=None):
        vals, saved = zip(*[w.testvalue(e, epos, mask=mask) for w in self.wf_factors])
        return np.prod(vals, axis=0), saved

    def gradient_value(self, e, epos):
        g, v, s = zip(*[w.gradient_value(e, epos) for w in self.wf_factors])
        return np.sum(g, axis=0), np.prod(v, axis=0), s

    def gradient_laplacian(self, e, epos):
        """multiplywf.py:121-129: lap = sum lap_i + 2 sum_{i<j} grad_i.grad_j."""
        g, l = zip(*[w.gradient_laplacian(e, epos) for w in self.wf_factors])
        cross = np.zeros(l[0].shape, dtype=self.dtype)
        for i in range(len(g)):
            for j in range(i + 1, len(g)):
                cross += np.sum(g[i] * g[j], axis=0)
        return np.sum(g, axis=0), np.sum(l, axis=0) + 2 * cross


class ThreeBodyJastrow(_ManyMixin):
    """Electron-electron-ion Jastrow e^U (``pyqmc/wf/three_body_jastrow.py:19-655``):

        U = 1/2 sum_e P_e,   P_e = sum_{j != e} sum_I sum_{klm} C_{Iklm,s(e,j)} a_k(r_eI) a_l(r_jI) b_m(r_ej),

    C = (c + c^T_{kl})/2 (:94-96), spin index s = [e down] + [j down].  Moving electron e changes U by
    P_e(new) - P_e(old) (testvalue :323-341); gradient_laplacian returns (grad U, lap U + |grad U|^2) (:541-655).
    Restated directly from these formulas (the reference spreads them over several einsum calls)."""

    def __init__(self, mol, a_basis, b_basis, rcut):
        self._nup, self._ndn = mol.nelec
        self._nelec = self._nup + self._ndn
        self.atoms = np.asarray(mol.atom_coords(), dtype=float)
        self.a_basis, self.b_basis, self.rcut = list(a_basis), list(b_basis), float(rcut)
        self.parameters = {"ccoeff": np.zeros((len(self.atoms), len(a_basis), len(a_basis), len(b_basis), 3))}
        self.dtype = float
        if hasattr(mol, "a"):  # minimal-image displacements (configs.dist, three_body_jastrow.py:70-92)
            from .pbc import minimal_image

            self._mi = minimal_image(mol.lattice_vectors())
        else:
            self._mi = lambda d: d

    def _C(self):
        c = self.parameters["ccoeff"]
        return 0.5 * (c + c.swapaxes(1, 2))

    def pgradient(self):
        """three_body_jastrow.py:657-719: dU/dc[I,k,l,m,sp] = 1/2 (X + X^T_kl), X = sum over pairs (i,j) of spin class sp of
        a_k(r_iI) a_l(r_jI) b_m(r_ij) (sp: up-up i<j, up-down, down-down i<j)."""
        x = self._x
        W, N = x.shape[:2]
        nup = self._nup
        dI = self._mi(x[:, :, None, :] - self.atoms[None, None])  # (W,N,A,3)
        a = jastrow_basis.evaluate(self.a_basis, self.rcut, dI, np.linalg.norm(dI, axis=-1), "value")  # (W,N,A,k)
        out = np.zeros((W, len(self.atoms), len(self.a_basis), len(self.a_basis), len(self.b_basis), 3))
        for sp, (ri, rj) in enumerate(((range(nup), range(nup)), (range(nup), range(nup, N)), (range(nup, N), range(nup, N)))):
            for i in ri:
                js = [j for j in rj if (j > i or sp == 1)]
                if not js:
                    continue
                d = self._mi(x[:, i, None, :] - x[:, js])
                b = jastrow_basis.evaluate(self.b_basis, self.rcut, d, np.linalg.norm(d, axis=-1), "value")  # (W,nj,m)
                out[..., sp] += np.einsum("wIk,wjIl,wjm->wIklm", a[:, i], a[:, js], b)
        return {"ccoeff": 0.5 * (out + out.swapaxes(2, 3))}

    def _terms(self, e, pos, xw, want):
        """P, grad P (3,...), lap P of electron e at pos (Wm,3) against walkers xw (Wm,N,3)."""
        others = np.arange(self._nelec) != e
        edown = int(e >= self._nup)
        xo = xw[:, others]  # (W, N-1, 3)
        sig = edown + (np.nonzero(others)[0] >= self._nup).astype(int)  # spin index per j
        C = self._C()[..., sig]  # (A,k,l,m,N-1)
        de_I = self._mi(pos[:, None, :] - self.atoms[None])  # (W,A,3)
        de_j = self._mi(pos[:, None, :] - xo)  # (W,N-1,3)
        dj_I = self._mi(xo[:, :, None, :] - self.atoms[None, None])  # (W,N-1,A,3)
        aj = jastrow_basis.evaluate(self.a_basis, self.rcut, dj_I, np.linalg.norm(dj_I, axis=-1), "value")  # (W,N-1,A,l)
        if want == "value":
            ae = jastrow_basis.evaluate(self.a_basis, self.rcut, de_I, np.linalg.norm(de_I, axis=-1), "value")
            b = jastrow_basis.evaluate(self.b_basis, self.rcut, de_j, np.linalg.norm(de_j, axis=-1), "value")
            return np.einsum("wIk,wjIl,wjm,Iklmj->w", ae, aj, b, C), None, None
        gae, ae = jastrow_basis.evaluate(self.a_basis, self.rcut, de_I, np.linalg.norm(de_I, axis=-1), "gradient_value")
        gb, b = jastrow_basis.evaluate(self.b_basis, self.rcut, de_j, np.linalg.norm(de_j, axis=-1), "gradient_value")
        P = np.einsum("wIk,wjIl,wjm,Iklmj->w", ae, aj, b, C)
        grad = np.einsum("wIkx,wjIl,wjm,Iklmj->xw", gae, aj, b, C) + np.einsum("wIk,wjIl,wjmx,Iklmj->xw", ae, aj, gb, C)
        if want == "gradient_value":
            return P, grad, None
        _, lae = jastrow_basis.evaluate(self.a_basis, self.rcut, de_I, np.linalg.norm(de_I, axis=-1), "gradient_laplacian")
        _, lb = jastrow_basis.evaluate(self.b_basis, self.rcut, de_j, np.linalg.norm(de_j, axis=-1), "gradient_laplacian")
        lap = (np.einsum("wIk,wjIl,wjm,Iklmj->w", lae, aj, b, C) + 2.0 * np.einsum("wIkx,wjIl,wjmx,Iklmj->w", gae, aj, gb, C)
               + np.einsum("wIk,wjIl,wjm,Iklmj->w", ae, aj, lb, C))
        return P, grad, lap

    def recompute(self, configs):
        self._x = configs.configs.copy()
        return self.value()

    def value(self):
        u = np.zeros(len(self._x))
        for e in range(self._nelec):
            u += 0.5 * self._terms(e, self._x[:, e], self._x, "value")[0]
        return np.ones(len(u)), u

    def testvalue(self, e, epos, mask=None):
        W = self._x.shape[0]
        mask = np.ones(W, dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        xw = self._x[mask]
        old = self._terms(e, xw[:, e], xw, "value")[0]
        x = epos.configs[mask]
        if x.ndim == 3:
            new = np.stack([self._terms(e, x[:, q], xw, "value")[0] for q in range(x.shape[1])], axis=1)
            return np.exp(new - old[:, None]), None
        return np.exp(self._terms(e, x, xw, "value")[0] - old), None

    def gradient_value(self, e, epos):
        P, grad, _ = self._terms(e, epos.configs, self._x, "gradient_value")
        old = self._terms(e, self._x[:, e], self._x, "value")[0]
        return grad, np.exp(P - old), None

    def gradient(self, e, epos):
        return self._terms(e, epos.configs, self._x, "gradient_value")[1]

    def gradient_laplacian(self, e, epos):
        _, grad, lap = self._terms(e, epos.configs, self._x, "gradient_laplacian")
        return grad, lap + np.sum(grad**2, axis=0)

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        mask = np.ones(len(self._x), dtype=bool) if mask is None else np.asarray(mask, dtype=bool)
        self._x[mask, e, :] = epos.configs[mask]
