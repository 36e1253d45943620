"""Wave-function objects with the reference's protocol, computed on the MI355X.

``Slater``, ``JastrowSpin`` and ``MultiplyWF`` present the methods of
``pyqmc/wf/slater.py:97-460``, ``pyqmc/wf/jastrowspin.py:20-419`` and
``pyqmc/wf/multiplywf.py:71-132`` (``recompute / value / gradient / gradient_value /
gradient_laplacian / testvalue / updateinternals`` with the same argument meaning and
return shapes) so drivers written against that protocol run unchanged.  Every number is
produced by the HIP library behind ``include/pyqmc_amd.h``; walker state (inverse
matrices, determinants, Jastrow sums, coordinates) stays on the device between calls and
only per-call inputs/outputs cross the boundary.
"""

import ctypes as C
import itertools

import numpy as np

from . import _ffi, func3d, tables
from .configs import OpenConfigs

_serial = itertools.count(1)


class DeviceWF:
    """Owner of one ``pqa_handle_t`` (one walker shard on one GPU)."""

    def __init__(self, mol, mo_coeff=None, determinants=None, a_basis=None, b_basis=None, device=0, tol=-1,
                 a3_basis=None, b3_basis=None, eval_gto_precision=None, image_rule="reference", twist_k=None):
        # what it takes to build this handle again (copy / pickle: the reference ships pickled wave functions to its worker
        # processes, mc.py:161, and its tests copy them, testwf.py:44): constructor arguments, parameters pushed since, walkers
        self._ctor = dict(mol=mol, mo_coeff=mo_coeff, determinants=determinants, a_basis=a_basis, b_basis=b_basis, device=device, tol=tol,
                          a3_basis=a3_basis, b3_basis=b3_basis, eval_gto_precision=eval_gto_precision, image_rule=image_rule,
                          twist_k=twist_k)
        self._pushed = {}
        self.mol = mol
        self.twisted = twist_k is not None and float(np.abs(twist_k).max()) > 1e-12
        self.nelec = tuple(int(n) for n in mol.nelec)
        self.N = sum(self.nelec)
        self.natom = int(mol.natm)
        self.has_slater = mo_coeff is not None
        self.cplx = False
        self.has_jastrow = a_basis is not None or b_basis is not None
        self.has_j3 = a3_basis is not None and b3_basis is not None
        keep = self._keep = {}  # host arrays referenced by the struct must outlive pqa_create
        s = _ffi.SystemStruct()
        s.natom, s.nelec_up, s.nelec_dn = self.natom, self.nelec[0], self.nelec[1]
        keep["xyz"] = _ffi.f64(mol.atom_coords())
        keep["chg"] = _ffi.f64(mol.atom_charges())
        s.atom_xyz = keep["xyz"].ctypes.data_as(_ffi.c_double_p)
        s.atom_charge = keep["chg"].ctypes.data_as(_ffi.c_double_p)

        def setd(name, arr):
            keep[name] = _ffi.f64(arr)
            setattr(s, name, keep[name].ctypes.data_as(_ffi.c_double_p))

        def seti(name, arr):
            keep[name] = np.ascontiguousarray(arr, dtype=np.int32)
            setattr(s, name, keep[name].ctypes.data_as(_ffi.c_int32_p))

        self.nmo = (0, 0)
        self.ndet, self.ndet_s = 1, (1, 1)
        if self.has_slater:
            bt = tables.basis_tables(mol)
            self.nao = bt["nao"]
            s.nshell, s.nprim, s.nao = len(bt["shell_l"]), len(bt["prim_exp"]), bt["nao"]
            for k in ("shell_atom", "shell_l", "shell_prim_off", "shell_ao_off"):
                seti(k, bt[k])
            setd("prim_exp", bt["prim_exp"])
            setd("prim_coef", bt["prim_coef"])
            coef, occ_up, occ_dn, dmap = tables.pack_determinants(self.nelec, determinants, tol)
            self.det_occup = [occ_up, occ_dn]
            self.det_map = dmap
            nmo = [int(o.max(initial=-1)) + 1 for o in (occ_up, occ_dn)]
            self.nmo = tuple(nmo)
            self.ndet, self.ndet_s = len(coef), (len(occ_up), len(occ_dn))
            self.cplx = any(np.iscomplexobj(np.asarray(m)) and np.abs(np.imag(m)).max(initial=0.0) > 0 for m in mo_coeff)
            dt = complex if self.cplx else float
            mo = [np.ascontiguousarray(np.asarray(mo_coeff[sp])[:, : nmo[sp]], dtype=dt) for sp in (0, 1)]
            self.mo_coeff = mo
            # complex orbitals travel as the real matrix [Re C | Im C]: the orbital kernel stays a real GEMM
            f = 2 if self.cplx else 1
            s.nmo_up, s.nmo_dn = f * nmo[0], f * nmo[1]
            s.complex_orbitals = int(self.cplx)
            setd("mo_up", self._mo_to_device(mo[0]))
            setd("mo_dn", self._mo_to_device(mo[1]))
            s.ndet, s.ndet_up, s.ndet_dn = self.ndet, len(occ_up), len(occ_dn)
            setd("det_coeff", coef)
            seti("det_occ_up", occ_up)
            seti("det_occ_dn", occ_dn)
            seti("det_map", dmap)
            self.det_coeff = coef
        s.has_slater = int(self.has_slater)
        self.na = self.nb = 0
        if self.has_jastrow:
            ak, ap, ra = tables.jastrow_basis_arrays(a_basis or [])
            bk, bp, rb = tables.jastrow_basis_arrays(b_basis or [])
            self.na, self.nb = len(ak), len(bk)
            s.na, s.nb, s.rcut_a, s.rcut_b = self.na, self.nb, ra, rb
            seti("a_kind", ak)
            setd("a_param", ap)
            seti("b_kind", bk)
            setd("b_param", bp)
            setd("acoeff", np.zeros((self.natom, self.na, 2)))
            setd("bcoeff", np.zeros((self.nb, 3)))
        self.na3 = self.nb3 = 0
        if self.has_j3:
            ak, ap, ra = tables.jastrow_basis_arrays(a3_basis)
            bk, bp, rb = tables.jastrow_basis_arrays(b3_basis)
            self.na3, self.nb3 = len(ak), len(bk)
            s.na3, s.nb3, s.rcut_a3, s.rcut_b3 = self.na3, self.nb3, ra, rb
            seti("a3_kind", ak)
            setd("a3_param", ap)
            seti("b3_kind", bk)
            setd("b3_param", bp)
            setd("ccoeff", np.zeros((self.natom, self.na3, self.na3, self.nb3, 3)))
        et = tables.ecp_tables(mol)
        self.necp = len(et["ecp_atom"])
        s.necp = self.necp
        for k in ("ecp_atom", "ecp_chan_off", "ecp_term_off", "ecp_term_n"):
            seti(k, et[k])
        setd("ecp_term_exp", et["ecp_term_exp"])
        setd("ecp_term_coef", et["ecp_term_coef"])
        self.pbc = hasattr(mol, "a")
        if self.twisted:
            if not (self.pbc and self.cplx):
                raise ValueError("a twist needs a periodic cell and complex orbitals")
            s.twisted = 1
            s.twist_k[:] = [float(v) for v in twist_k]
            self.lattice = np.asarray(mol.lattice_vectors(), dtype=float)
        if self.pbc:
            from .configs import MinimalImageDistance

            lat = np.asarray(mol.lattice_vectors(), dtype=float)
            s.pbc = 2 if MinimalImageDistance(lat).kind == "general" else 1
            s.lattice[:] = list(lat.ravel())
            if self.has_slater:
                from . import pbc as _pbc

                pt = _pbc.periodic_tables(mol, eval_gto_precision, image_rule=image_rule)
                s.nL = len(pt["Ls"])
                setd("Ls", pt["Ls"])
                seti("num_Ls", pt["num_Ls"])
                setd("atom_cut", pt["atom_cut"])
                setd("shell_cut", pt["shell_cut"])
                if pt["member"] is not None:
                    s.lattice_prim[:] = list(np.asarray(pt["lattice_prim"], dtype=float).ravel())
                    seti("img_n", pt["img_n"])
                    seti("atom_n", pt["atom_n"])
                    seti("member_class", pt["member_class"])
                    keep["member"] = np.ascontiguousarray(pt["member"], dtype=np.uint8)
                    s.member = keep["member"].ctypes.data_as(C.POINTER(C.c_uint8))
                    s.member_M, s.n_member_class = int(pt["member_M"]), int(pt["member"].shape[0])
        self._struct = s
        self._h = C.c_void_p()
        self._zero = [None, None]  # per spin: did the last pqa_wf_update leave a vanished determinant (None: unknown)
        lib = _ffi.lib()
        rc = lib.pqa_create(C.byref(s), int(device), C.byref(self._h))
        if rc != 0:
            msg = lib.pqa_last_error(None)
            raise _ffi.PqaError(f"pqa_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.device = int(device)
        self.W = 0
        self._ewald_key = None
        if self.pbc:
            self.set_ewald()

    def __del__(self):
        h = getattr(self, "_h", None)
        if h is not None and h.value:
            try:
                _ffi.lib().pqa_destroy(h)
            except Exception:
                pass
            self._h = None

    # ------------------------------------------------------------------
    def call(self, name, *args):
        if name != "pqa_wf_eval":  # (anything else may change the determinants: the flag pqa_wf_update returned no longer describes them)
            self._zero = [None, None]
        _ffi.check(self._h, getattr(_ffi.lib(), name)(self._h, *args))

    def call_int(self, name, *args):
        return int(getattr(_ffi.lib(), name)(self._h, *args))

    def _mo_to_device(self, m):
        m = np.asarray(m)
        return np.ascontiguousarray(np.concatenate([m.real, m.imag], axis=1) if self.cplx else m, dtype=float)

    @property
    def cdtype(self):
        return complex if self.cplx else float

    def set_param(self, name, value):
        self._pushed[name] = np.array(value, copy=True)
        a = self._mo_to_device(value) if name.startswith("mo_coeff") else _ffi.f64(value)
        self.call("pqa_set_param", name.encode(), _ffi.ptr(a), a.size)

    # ---- copy / pickle: a handle cannot be aliased or serialised, it is REBUILT ------------------------------
    def __getstate__(self):
        """Host-side recipe of the handle: tables (constructor arguments), parameters pushed since, resident walkers.
        Device state (inverses, caches, Jastrow sums) is recomputed from the walkers on the other side."""
        return {"ctor": self._ctor, "pushed": self._pushed, "configs": self.configs() if self.W else None, "ewald": self._ewald_key}

    def __setstate__(self, st):
        self.__init__(**st["ctor"])
        for name, value in st["pushed"].items():
            self.set_param(name, value)
        if st["ewald"] is not None and st["ewald"] != self._ewald_key:
            self.set_ewald(*st["ewald"])
        if st["configs"] is not None:
            self.recompute(st["configs"])

    def __copy__(self):  # a shallow copy would share one device handle between two "independent" wave functions
        import copy

        return copy.deepcopy(self)

    # fused device-resident entry points ---------------------------------
    def set_ewald(self, ewald_gmax=200, nlatvec=1):
        """Upload the Ewald tables (``Ewald(cell, ewald_gmax, nlatvec)``, observables/ewald.py:95-107)."""
        if self._ewald_key == (ewald_gmax, nlatvec):
            return
        from .ewald import ewald_tables

        t = ewald_tables(self.mol, ewald_gmax, nlatvec)
        self.call("pqa_set_ewald", t["alpha"], len(t["gweight"]), _ffi.ptr(t["gpoints"]), _ffi.ptr(t["gweight"]),
                  _ffi.ptr(t["ion_cos"]), _ffi.ptr(t["ion_sin"]), t["ee_const"], t["ei_const"], t["ii"], _ffi.ptr(t["gidx"]),
                  _ffi.ptr(t["recip"]))
        self._ewald_key = (ewald_gmax, nlatvec)
        self.ewald = t

    def wrap_delta(self):
        """(W, nelec, 3) wraps accumulated by the last fused sweep call."""
        out = np.zeros((self.W, self.N, 3), dtype=np.int32)
        self.call("pqa_get_wrap", _ffi.ptr(out))
        return out

    def recompute(self, configs):
        x = _ffi.f64(configs)
        W = x.shape[0]
        sign, logv = np.empty(W, dtype=self.cdtype), np.empty(W)
        self.call("pqa_wf_recompute", _ffi.ptr(x), W, _ffi.ptr(sign), _ffi.ptr(logv))
        self.W = W
        return sign, logv

    def value(self):
        sign, logv = np.empty(self.W, dtype=self.cdtype), np.empty(self.W)
        self.call("pqa_wf_value", _ffi.ptr(sign), _ffi.ptr(logv))
        return sign, logv

    def configs(self):
        out = np.empty((self.W, self.N, 3))
        self.call("pqa_get_configs", _ffi.ptr(out))
        return out

    def energy(self, threshold=10.0, rot=None, unif=None, seed=0):
        """(6, W): ke, ee, ei, ecp, grad2, total of the resident walkers; complex (ecp and total carry an imaginary
        part, eval_ecp.py:89) when the orbitals are."""
        out = np.empty((7 if self.cplx else 6, self.W))
        rot = None if rot is None else _ffi.f64(rot)
        unif = None if unif is None else _ffi.f64(unif)
        self.call("pqa_energy", float(threshold), _ffi.ptr(rot), _ffi.ptr(unif), int(seed), _ffi.ptr(out))
        return self._complex_energy(out, 0) if self.cplx else out

    @staticmethod
    def _complex_energy(a, axis):
        """rows/columns (ke, ee, ei, Re ecp, grad2, Re total, Im ecp) -> six complex entries (Im total = Im ecp)."""
        a = np.moveaxis(a, axis, 0)
        out = a[:6].astype(complex)
        out[3] += 1j * a[6]
        out[5] += 1j * a[6]
        return np.moveaxis(out, 0, axis)

    def vmc_sweeps(self, tstep, nsteps, gauss=None, unif=None, threshold=10.0, ecp_rot=None, ecp_unif=None, seed=0,
                   energy=True, record=False):
        acc = np.empty(nsteps)
        en = np.empty((nsteps, 7 if self.cplx else 6)) if energy else None
        rec = np.empty((nsteps, self.N, self.W), dtype=np.uint8) if record else None
        g = None if gauss is None else _ffi.f64(gauss)
        u = None if unif is None else _ffi.f64(unif)
        er = None if ecp_rot is None else _ffi.f64(ecp_rot)
        eu = None if ecp_unif is None else _ffi.f64(ecp_unif)
        self.call("pqa_vmc_sweeps", float(tstep), int(nsteps), _ffi.ptr(g), _ffi.ptr(u), float(threshold), _ffi.ptr(er),
                  _ffi.ptr(eu), int(seed), _ffi.ptr(acc), _ffi.ptr(en), _ffi.ptr(rec))
        if energy and self.cplx:
            en = self._complex_energy(en, 1)
        return acc, en, (rec.astype(bool) if record else None)

    def philox_tapes(self, seed, nsteps, W):
        """The draws ``vmc_sweeps(..., seed=seed)`` makes for walkers 0..W-1: gauss (nsteps,N,W,3), unif (nsteps,N,W)."""
        gauss, unif = np.empty((nsteps, self.N, W, 3)), np.empty((nsteps, self.N, W))
        for s in range(nsteps):
            self.call("pqa_philox_tapes", int(seed), s, int(W), _ffi.ptr(gauss[s]), _ffi.ptr(unif[s]))
        return gauss, unif

    def philox_dmc_tapes(self, seed, nsteps, W, tmoves=True):
        """The draws ``dmc_steps(..., tapes=None, seed=seed)`` makes for walkers 0..W-1, as the tape dictionary ``dmc_steps`` takes
        (``pqa_philox_dmc_tapes``): lets the CPU oracle replay a device-RNG DMC block."""
        N, necp = self.N, getattr(self, "necp", 0)
        t = {"gauss": np.empty((nsteps, N, W, 3)), "unif": np.empty((nsteps, N, W))}
        if necp:
            t["ecp_rot"], t["ecp_unif"] = np.empty((nsteps + 1, N, necp, 3, 3)), np.empty((nsteps + 1, N, necp, W))
            if tmoves:
                t["tm_rot"], t["tm_unif"] = np.empty((nsteps, N, necp, 3, 3)), np.empty((nsteps, N, necp, W))
                t["tm_u1"], t["tm_u2"] = np.empty((nsteps, N, W)), np.empty((nsteps, N, W))
        tp = _ffi.DmcTapes()
        for name, _ in _ffi.DmcTapes._fields_:
            if name in t:
                setattr(tp, name, t[name].ctypes.data)
        self.call("pqa_philox_dmc_tapes", int(seed), int(nsteps), int(W), C.addressof(tp))
        return t

    def resample(self, newinds):
        """``pqa_resample``: walker w of the resident state becomes a copy of walker ``newinds[w]``."""
        idx = np.ascontiguousarray(newinds, dtype=np.int32)
        assert idx.shape == (self.W,)
        self.call("pqa_resample", _ffi.ptr(idx))

    def get_walkers(self, idx, out=None):
        """``pqa_get_walkers``: coordinates (n,N,3) of the resident walkers ``idx``; ``out``: raw pointer (int) of a device or
        host buffer to fill instead of returning a new host array."""
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        if out is not None:
            self.call("pqa_get_walkers", _ffi.ptr(idx), len(idx), C.c_void_p(out))
            return None
        res = np.empty((len(idx), self.N, 3))
        self.call("pqa_get_walkers", _ffi.ptr(idx), len(idx), _ffi.ptr(res))
        return res

    def branch_exchange(self, keep_src, recv_x, nrecv):
        """``pqa_branch_exchange``: new ensemble = resident walkers ``keep_src`` (state gathered) + ``nrecv`` received walkers
        (``recv_x``: host array or raw device pointer), whose state alone is recomputed."""
        keep = np.ascontiguousarray(keep_src, dtype=np.int32)
        ptr = C.c_void_p(recv_x) if isinstance(recv_x, int) else _ffi.ptr(None if recv_x is None else _ffi.f64(recv_x))
        self.call("pqa_branch_exchange", _ffi.ptr(keep), len(keep), ptr, int(nrecv))

    def dmc_steps(self, tstep, nsteps, weights, branchcut, e_trial, e_est, threshold=10.0, tapes=None, seed=0, cont=False):
        """``pqa_dmc_steps``: ``nsteps`` DMC steps on the resident walkers.  ``weights`` (W) is updated in place.
        ``tapes``: dict of the replay arrays of ``pqa_dmc_tapes_t`` or None (device Philox streams).
        Returns (step_avg (nsteps,7) — (nsteps,8) for complex wave functions —, step_acc (nsteps,2))."""
        assert weights.dtype == np.float64 and weights.flags.c_contiguous and weights.shape == (self.W,)
        avg, acc = np.empty((nsteps, 8 if self.cplx else 7)), np.empty((nsteps, 2))  # complex: + the weighted mean of Im ecp
        tp, keep = None, []
        if tapes is not None:
            tp = _ffi.DmcTapes()
            for name, _ in _ffi.DmcTapes._fields_:
                a = tapes.get(name)
                if a is not None:
                    a = _ffi.f64(a)
                    keep.append(a)
                    setattr(tp, name, a.ctypes.data)
        if cont:  # start from the energies the previous call ended with (pqa_dmc_continue)
            self.call("pqa_dmc_continue", 1)
        self.call("pqa_dmc_steps", float(tstep), int(nsteps), float(branchcut), float(e_trial), float(e_est), float(threshold),
                  _ffi.ptr(weights), None if tp is None else C.addressof(tp), int(seed), _ffi.ptr(avg), _ffi.ptr(acc))
        return avg, acc

    def dmc_can_continue(self):
        """True while the device still holds the energies its last ``dmc_steps`` call ended with for the resident state (``pqa_dmc_can_continue``)."""
        return bool(self.call_int("pqa_dmc_can_continue"))

    # measurement ----------------------------------------------------------
    def sync(self):
        self.call("pqa_sync")

    def timer_start(self):
        self.call("pqa_timer_start")

    def timer_stop(self):
        ms = C.c_double()
        self.call("pqa_timer_stop", C.byref(ms))
        return ms.value

    def profile_enable(self, on=True):
        self.call("pqa_profile_enable", int(bool(on)))

    def profile_query(self):
        n, ms, pc = C.c_int64(), C.c_double(), C.c_double()
        self.call("pqa_profile_query", C.byref(n), C.byref(ms), C.byref(pc))
        return n.value, ms.value, pc.value

    def profile_query_commit(self):
        """(launches, total ms) of the Sherman-Morrison flush kernel (``k_flush_lw``) since ``profile_enable``."""
        n, ms = C.c_int64(), C.c_double()
        self.call("pqa_profile_query_commit", C.byref(n), C.byref(ms))
        return n.value, ms.value

    def profile_query_part(self):
        """(launches, total ms, partial-sum groups per walker) of the bracketed k_move_part_lw launches."""
        n, ms, g = C.c_int64(), C.c_double(), C.c_int()
        self.call("pqa_profile_query_part", C.byref(n), C.byref(ms), C.byref(g))
        return n.value, ms.value, g.value

    def set_ecp_naip(self, naip):
        """Quadrature rule of the energy pass's ECP integrator (pqa_set_ecp_naip): 6/12/18/26/32/50, None = per-atom default."""
        naip = 0 if naip is None else int(naip)
        if getattr(self, "_naip", 0) != naip:
            self.call("pqa_set_ecp_naip", naip)
            self._naip = naip

    def set_ecp_batched(self, naip, nsd=0, nsr=0):
        """Switch the energy pass's ECP part to the batched integrator (pqa_set_ecp_batched; naip per ECP atom) or, with None, back."""
        if naip is None:
            self.call("pqa_set_ecp_batched", 0, None, 0, 0)
            return
        a = np.ascontiguousarray(naip, dtype=np.int32)
        if a.shape != (self.necp,):
            raise ValueError(f"naip: one entry per ECP atom ({self.necp})")
        self.call("pqa_set_ecp_batched", 1, _ffi.ptr(a), int(nsd), int(nsr))

    def ecp_batched_moves(self, e, tau, rot, unif=None, seed=0):
        """(weight (W, P), pos (W, P, 3)) of pqa_ecp_batched_moves for electron e."""
        P = self.call_int("pqa_ecp_batched_nselected")
        weight, pos = np.empty((self.W, P)), np.empty((self.W, P, 3))
        rot = _ffi.f64(rot)
        unif = None if unif is None or unif.size == 0 else _ffi.f64(unif)
        self.call("pqa_ecp_batched_moves", int(e), float(tau), _ffi.ptr(rot), _ffi.ptr(unif), int(seed), _ffi.ptr(weight), _ffi.ptr(pos))
        return weight, pos

    # product wave function in one call per protocol method (pqa_wf_eval / pqa_wf_update) -------------------------
    def wf_eval(self, e, pts, jmode, keep):
        """(9, W): Slater ratio rows (value, gradient, laplacian) + Jastrow rows (gradient, value | laplacian) at pts (W, 3)."""
        out = np.empty((9, self.W))
        self.call("pqa_wf_eval", int(e), _ffi.ptr(pts), int(jmode), int(keep), _ffi.ptr(out))
        return out

    def wf_update(self, e, x, m8, use_saved, spin):
        flag = C.c_int()
        self.call("pqa_wf_update", int(e), _ffi.ptr(x), _ffi.ptr(m8), int(use_saved), C.byref(flag))
        self._zero[spin] = bool(flag.value)

    def last_ecp_points(self):
        n = C.c_int64()
        self.call("pqa_last_ecp_points", C.byref(n))
        return n.value

    def eval_ao(self, pts, ncomp):
        p = _ffi.f64(pts).reshape(-1, 3)
        out = np.empty((ncomp, p.shape[0], self.nao))
        self.call("pqa_eval_ao", _ffi.ptr(p), p.shape[0], ncomp, _ffi.ptr(out))
        return out

    def eval_mo(self, spin, pts, ncomp, use_mfma=True):
        p = _ffi.f64(pts).reshape(-1, 3)
        f = 2 if self.cplx else 1
        out = np.empty((ncomp, p.shape[0], f * self.nmo[spin]))
        self.call("pqa_eval_mo", int(spin), _ffi.ptr(p), p.shape[0], ncomp, int(use_mfma), _ffi.ptr(out))
        return out[..., : self.nmo[spin]] + 1j * out[..., self.nmo[spin] :] if self.cplx else out


class _DeviceParams(dict):
    """``wf.parameters``: a dict of host arrays whose assignments are pushed to the device."""

    def __init__(self, dev, items, to_device=None):
        super().__init__(items)
        self._dev = dev
        self._to_device = to_device or {}  # key -> map from the parameter's public layout to the device's

    def _send(self, key, value):
        f = self._to_device.get(key)
        self._dev.set_param(key, value if f is None else f(value))

    def __setitem__(self, key, value):
        value = np.array(value)
        if key in self:
            have = np.asarray(self[key])
            if value.shape != have.shape:
                raise ValueError(f"parameter {key} has shape {have.shape}, got {value.shape}")
            if np.iscomplexobj(value) and not np.iscomplexobj(have):
                if np.any(value.imag != 0):
                    raise TypeError(f"parameter {key} is real on the device; refusing to drop the imaginary part of a complex value")
                value = value.real
            value = value.astype(have.dtype)
        else:
            value = value.astype(complex if np.iscomplexobj(value) else float)
        super().__setitem__(key, value)
        self._send(key, value)

    def push(self):
        for k, v in self.items():
            self._send(k, v)

    def __reduce__(self):  # the device binding (and the closures of _to_device) do not travel: the owning factor re-wraps
        return (dict, (dict(self),))


class _DeviceFactor:
    """Copy / pickle behaviour shared by the factors: the state is every attribute plus the parameter VALUES; the shared
    DeviceWF is rebuilt once per copy (``copy.deepcopy`` / ``pickle`` memoise it), and the parameters are bound to it again."""

    def _bind_parameters(self, items):
        self.parameters = _DeviceParams(self._dev, items)

    def __getstate__(self):
        d = dict(self.__dict__)
        d["parameters"] = {k: np.array(v, copy=True) for k, v in self.parameters.items()}
        return d

    def __setstate__(self, d):
        items = d.pop("parameters")
        self.__dict__.update(d)
        self._bind_parameters(items)

    def __copy__(self):
        import copy

        return copy.deepcopy(self)


def _mask_args(mask, W):
    """-> (bool mask or None, uint8 array or None)."""
    if mask is None:
        return None, None
    m = np.asarray(mask, dtype=bool)
    if m.shape != (W,):
        raise ValueError("mask must have one entry per walker")
    return m, np.ascontiguousarray(m, dtype=np.uint8)


def _xyz(dev, obj):
    """Coordinates handed to the device: the container's folded positions, or — for a twisted handle, which derives the
    wrap phase from the position itself (include/pyqmc_amd.h) — the unfolded ones ``configs + wrap @ lattice``."""
    x = np.asarray(obj.configs, dtype=float)
    if getattr(dev, "twisted", False):
        x = x + np.asarray(obj.wrap, dtype=float) @ dev.lattice
    return x


def _points(epos, mask, dev=None):
    """-> (pts (nrow,npt,3), widx int32 or None, aux?)"""
    x = _xyz(dev, epos)
    aux = x.ndim == 3
    widx = None
    if mask is not None:
        widx = np.ascontiguousarray(np.nonzero(mask)[0], dtype=np.int32)
        x = x[mask]
    npt = x.shape[1] if aux else 1
    pts = np.ascontiguousarray(x.reshape(x.shape[0], npt, 3))
    return pts, widx, aux


def _testvalue_many(dev, factors, e, epos, mask):
    """(nrow, len(e)) ratios Psi(e_i -> epos)/Psi for ONE auxiliary position per walker (``testvalue_many``)."""
    es = np.ascontiguousarray(np.atleast_1d(e), dtype=np.int32)
    x = _xyz(dev, epos)
    if x.ndim != 2:
        raise ValueError("testvalue_many takes one position per walker: epos.configs (nconf, 3)")
    widx = None
    if mask is not None:
        mask = np.asarray(mask, dtype=bool)
        widx = np.ascontiguousarray(np.nonzero(mask)[0], dtype=np.int32)
        x = x[mask]
    pts = np.ascontiguousarray(x)
    cplx = bool(getattr(dev, "cplx", False)) and bool(factors & 1)  # complex determinant ratios; Jastrow factors stay real
    out = np.empty((len(pts), len(es)), dtype=complex if getattr(dev, "cplx", False) else float)
    if len(pts) and len(es):
        dev.call("pqa_testvalue_many", _ffi.ptr(es), len(es), _ffi.ptr(pts), len(pts), _ffi.ptr(widx), int(factors), _ffi.ptr(out))
    return out if cplx else np.real(out)


def orbital_inputs(mol, mf, determinants=None, with_fold=False):
    """(mol, mo_coeff (2)[nao, nmo], determinants) the device is built from — the role of
    ``pyscftools.orbital_evaluator_from_pyscf`` (pyscftools.py:105-191).  Open systems pass through.  A periodic
    ``mol`` (a ``pyqmc_amd.pbc.get_supercell`` result, or a plain cell = supercell with S = 1) takes a k-point mean
    field (``kpts``, ``mo_coeff[s][k]``, ``mo_occ[s][k]``): the orbitals of the k-points that fold onto the
    supercell's Gamma point are truncated to the used columns per k (:170-173), determinants given per k are
    flattened onto the k-concatenated MO list (determinant_tools.py:91-104) and the Bloch phases are folded into
    real supercell coefficients (``pbc.fold_mo_coeff``)."""
    mf = mf.to_uhf() if hasattr(mf, "to_uhf") else mf
    if not hasattr(mol, "a"):
        return (mol, mf.mo_coeff, determinants, None) + ((None,) if with_fold else ())
    from . import pbc as _pbc

    if not hasattr(mol, "original_cell"):
        mol = _pbc.get_supercell(mol, np.eye(3))
    kpts = np.asarray(mf.kpts, dtype=float).reshape(-1, 3)
    want = _pbc.get_supercell_kpts(mol)
    rec = mol.original_cell.reciprocal_vectors()

    def same(k1, k2):
        d = (k1 - k2) @ np.linalg.inv(rec)
        return np.abs(d - np.round(d)).max() < 1e-9

    twist_k = _pbc.common_twist(mol, kpts)
    if twist_k is not None and len(kpts) == mol.scale:  # the mean field holds exactly the k-points of one (non-zero) twist
        want = want + twist_k
    kinds = [next((i for i, k in enumerate(kpts) if same(k, w)), None) for w in want]
    if any(i is None for i in kinds):
        raise ValueError(f"the mean field lacks some of the {len(want)} k-points that fold onto the supercell (pyscftools.py:161-166)")
    if not (twist_k is not None and len(kpts) == mol.scale):
        twist_k = None
    if determinants is None:
        determinants = [(1.0, [[list(np.nonzero(np.asarray(o) > 0.5)[0]) for o in mf.mo_occ[sp]] for sp in (0, 1)])]
    if len(determinants[0][1][0]) and hasattr(determinants[0][1][0][0], "__len__"):  # per-k occupations
        max_orb = np.amax([[[int(np.max(k, initial=-1)) + 1 for k in sp] for sp in det] for _, det in determinants], axis=0)
        offs = np.pad(np.cumsum(max_orb[:, kinds], axis=1)[:, :-1], ((0, 0), (1, 0)))
        flat = [(wt, [[int(i) + int(offs[sp][ki]) for ki, k in enumerate(kinds) for i in det[sp][k]] for sp in (0, 1)])
                for wt, det in determinants]
        mo = [[np.asarray(mf.mo_coeff[sp][k])[:, : max_orb[sp][k]] for k in kinds] for sp in (0, 1)]
    else:  # already flat: indices into the concatenation of the full per-k blocks
        flat = determinants
        mo = [[np.asarray(mf.mo_coeff[sp][k]) for k in kinds] for sp in (0, 1)]
    out = (mol, _pbc.fold_mo_coeff(mol, kpts[kinds], mo), flat, twist_k)
    # with_fold: what maps the reference's per-k parameter layout (nao_prim, sum_k nmo_k) to the folded matrices and back
    return out + (({"kpts": kpts[kinds], "blocks": mo, "nmo_k": [[b.shape[1] for b in mo[sp]] for sp in (0, 1)]},) if with_fold else ())


class Slater(_DeviceFactor):
    """Multi-determinant Slater factor (protocol of ``pyqmc/wf/slater.py:97-460``).

    ``mol``/``mf`` are the duck-typed containers of ``pyqmc_amd.systems`` (or PySCF-like
    objects exposing the same attributes); ``determinants`` is the list format of the
    reference's ``determinants=`` argument (slater.py:166-180)."""

    def __init__(self, mol, mf, determinants=None, tol=None, device=0, eval_gto_precision=None, image_rule="reference",
                 _dev=None, _fold=None):
        if _dev is None:
            mol, mo_coeff, determinants, twist_k, _fold = orbital_inputs(mol, mf, determinants, with_fold=True)
            _dev = DeviceWF(mol, mo_coeff=mo_coeff, determinants=determinants, device=device,
                            tol=-1 if tol is None else tol, eval_gto_precision=eval_gto_precision, image_rule=image_rule,
                            twist_k=twist_k)
        self._mol = mol
        self._nelec = tuple(mol.nelec)
        self._dev = _dev
        self._fold = _fold
        items = {"det_coeff": _dev.det_coeff.copy(), "mo_coeff_alpha": _dev.mo_coeff[0].copy(), "mo_coeff_beta": _dev.mo_coeff[1].copy()}
        if self._fold is not None:
            for sp, key in enumerate(("mo_coeff_alpha", "mo_coeff_beta")):
                items[key] = np.concatenate([np.asarray(b) if _dev.cplx else np.real(b) for b in self._fold["blocks"][sp]], axis=1)
        self._bind_parameters(items)
        self._det_occup = [o.tolist() for o in _dev.det_occup]
        self._det_map = _dev.det_map
        self.dtype = _dev.cdtype  # slater.py:212-216
        self._saved = None

    def _bind_parameters(self, items):
        to_device = {}
        if self._fold is not None:
            # periodic determinants (real or complex): the parameters have the reference's layout — per-k blocks (nao_prim, nmo_k)
            # concatenated over k (orbitals.py:154-160) — and are folded into supercell coefficients when pushed
            from . import pbc as _pbc

            cplx = self._dev.cplx
            for sp, key in enumerate(("mo_coeff_alpha", "mo_coeff_beta")):
                split = np.cumsum(self._fold["nmo_k"][sp])[:-1]
                to_device[key] = (lambda v, split=split: (np.asarray if cplx else np.real)(_pbc.fold_mo_coeff(
                    self._mol, self._fold["kpts"], [np.split(np.asarray(v), split, axis=1)] * 2)[0]))
        self.parameters = _DeviceParams(self._dev, items, to_device)

    def _spin(self, e):
        return int(e >= self._nelec[0])

    def recompute(self, configs):
        self.parameters.push()
        x = _ffi.f64(_xyz(self._dev, configs))
        W = x.shape[0]
        sign, logv = np.empty(W, dtype=self._dev.cdtype), np.empty(W)
        self._dev.call("pqa_slater_recompute", _ffi.ptr(x), W, _ffi.ptr(sign), _ffi.ptr(logv))
        self._dev.W = W
        return sign, logv

    def value(self):
        W = self._dev.W
        sign, logv = np.empty(W, dtype=self._dev.cdtype), np.empty(W)
        self._dev.call("pqa_slater_value", _ffi.ptr(sign), _ffi.ptr(logv))
        return sign, logv

    def _ratios(self, e, epos, mask, ncomp, keep):
        m, _ = _mask_args(mask, self._dev.W)
        pts, widx, aux = _points(epos, m, self._dev)
        nrow, npt = pts.shape[0], pts.shape[1]
        out = np.empty((ncomp, nrow * npt), dtype=self._dev.cdtype)
        if nrow:
            self._dev.call("pqa_slater_eval", int(e), _ffi.ptr(pts), nrow, npt, _ffi.ptr(widx), ncomp, int(keep), _ffi.ptr(out))
        return out, nrow, npt, aux

    def gradient_value(self, e, epos):
        r, *_ = self._ratios(e, epos, None, 5, True)
        with np.errstate(divide="ignore", invalid="ignore"):
            deriv = r[1:4] / r[0]
        deriv[~np.isfinite(deriv)] = 0.0
        val = r[0].copy()
        val[~np.isfinite(val)] = 1.0
        self._saved = ("pqa-slater-saved", int(e), next(_serial))
        return deriv, val, self._saved

    def gradient(self, e, epos):
        r, *_ = self._ratios(e, epos, None, 5, False)
        with np.errstate(divide="ignore", invalid="ignore"):
            return r[1:4] / r[0]

    def gradient_laplacian(self, e, epos):
        r, *_ = self._ratios(e, epos, None, 5, False)
        with np.errstate(divide="ignore", invalid="ignore"):
            r = r / r[:1]
        return r[1:4], r[4]

    def testvalue(self, e, epos, mask=None):
        r, nrow, npt, aux = self._ratios(e, epos, mask, 1, False)
        r = r[0].reshape(nrow, npt) if aux else r[0]
        return r, None

    def testvalue_many(self, e, epos, mask=None):
        """slater.py:448-460: ratios for moving each electron of ``e`` to ``epos`` -> (nconf[mask], len(e))."""
        return _testvalue_many(self._dev, 1, e, epos, mask)

    def pgradient(self):
        """slater.py:462-542: d Psi / Psi w.r.t. ``det_coeff`` (nconf, ndet) and the orbital coefficients
        (nconf, nao, nmo_s); zero-sized entries are dropped like the reference does (:537-541).  Complex determinants
        (complex Bloch phases or coefficients, twisted cells) give complex, holomorphic derivatives."""
        d = self._dev
        W = d.W
        dt = complex if d.cplx else float
        out = {"det_coeff": np.empty((W, d.ndet), dtype=dt), "mo_coeff_alpha": np.empty((W, d.nao, d.nmo[0]), dtype=dt),
               "mo_coeff_beta": np.empty((W, d.nao, d.nmo[1]), dtype=dt)}
        d.call("pqa_slater_pgradient", _ffi.ptr(out["det_coeff"]),
               _ffi.ptr(out["mo_coeff_alpha"]) if out["mo_coeff_alpha"].size else None,
               _ffi.ptr(out["mo_coeff_beta"]) if out["mo_coeff_beta"].size else None)
        if d.pbc and self._fold is not None:  # periodic: chain rule back to the per-k blocks of the parameter (slater.py:511-527)
            from . import pbc as _pbc

            for sp, key in enumerate(("mo_coeff_alpha", "mo_coeff_beta")):
                if out[key].size:
                    out[key] = _pbc.unfold_mo_gradient(self._mol, self._fold["kpts"], out[key], self._fold["nmo_k"][sp])
        return {k: v for k, v in out.items() if v.size}

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        s = self._spin(e)
        flag = C.c_int()
        self._dev.call("pqa_slater_has_zero", s, C.byref(flag))
        if flag.value:  # slater.py:269-275
            import warnings

            warnings.warn("Found a zero in the wave function. Recomputing everything. This should not happen often.")
            self.recompute(configs)
            return
        _, m8 = _mask_args(mask, self._dev.W)
        x = _ffi.f64(_xyz(self._dev, epos))
        use_saved = saved_values is not None and saved_values is self._saved and saved_values[1] == int(e)
        self._dev.call("pqa_slater_update", int(e), _ffi.ptr(x), _ffi.ptr(m8), int(use_saved))
        self._saved = None

    # test access to internals, in the reference's layout
    def _get_state(self, s):
        W, D, n = self._dev.W, self._dev.ndet_s[s], self._nelec[s]
        if self._dev.cplx:  # phases complex (W,D) followed by logs (W,D)
            inv, raw = np.empty((W, D, n, n), dtype=complex), np.empty(3 * W * D)
            self._dev.call("pqa_slater_get_state", s, _ffi.ptr(inv), _ffi.ptr(raw))
            ph = raw[: 2 * W * D].reshape(W, D, 2)
            return inv, np.array([ph[..., 0] + 1j * ph[..., 1], raw[2 * W * D :].reshape(W, D)])
        inv, dets = np.empty((W, D, n, n)), np.empty((2, W, D))
        self._dev.call("pqa_slater_get_state", s, _ffi.ptr(inv), _ffi.ptr(dets))
        return inv, dets


class JastrowSpin(_DeviceFactor):
    """One- and two-body Jastrow factor (protocol of ``pyqmc/wf/jastrowspin.py:20-419``).
    ``a_basis``/``b_basis``: lists of ``pyqmc_amd.func3d`` descriptors."""

    def __init__(self, mol, a_basis, b_basis, device=0, _dev=None):
        self._mol = mol
        self._nelec = int(np.sum(mol.nelec))
        if _dev is None:
            _dev = DeviceWF(mol, a_basis=list(a_basis), b_basis=list(b_basis), device=device)
        self._dev = _dev
        self._bind_parameters({"bcoeff": np.zeros((_dev.nb, 3)), "acoeff": np.zeros((_dev.natom, _dev.na, 2))})
        self.dtype = float

    def recompute(self, configs):
        self.parameters.push()
        x = _ffi.f64(_xyz(self._dev, configs))
        W = x.shape[0]
        u = np.empty(W)
        self._dev.call("pqa_jastrow_recompute", _ffi.ptr(x), W, _ffi.ptr(u))
        self._dev.W = W
        return np.ones(W), u

    def value(self):
        u = np.empty(self._dev.W)
        self._dev.call("pqa_jastrow_value", _ffi.ptr(u))
        return np.ones(len(u)), u

    def _eval(self, e, epos, mask, mode):
        m, _ = _mask_args(mask, self._dev.W)
        pts, widx, aux = _points(epos, m, self._dev)
        nrow, npt = pts.shape[0], pts.shape[1]
        out = np.empty(nrow * npt) if mode == 0 else np.empty((4, nrow))
        if nrow:
            self._dev.call("pqa_jastrow_eval", int(e), _ffi.ptr(pts), nrow, npt, _ffi.ptr(widx), mode, _ffi.ptr(out))
        return out, nrow, npt, aux

    def testvalue(self, e, epos, mask=None):
        r, nrow, npt, aux = self._eval(e, epos, mask, 0)
        return (r.reshape(nrow, npt) if aux else r), None

    def testvalue_many(self, e, epos, mask=None):
        """jastrowspin.py:421-455 -> (nconf[mask], len(e))."""
        return _testvalue_many(self._dev, 2, e, epos, mask)

    def gradient_value(self, e, epos):
        r, *_ = self._eval(e, epos, None, 1)
        return r[:3], r[3], None

    def gradient(self, e, epos):
        return self._eval(e, epos, None, 1)[0][:3]

    def gradient_laplacian(self, e, epos):
        r, *_ = self._eval(e, epos, None, 2)
        return r[:3], r[3]

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        _, m8 = _mask_args(mask, self._dev.W)
        x = _ffi.f64(_xyz(self._dev, epos))
        self._dev.call("pqa_jastrow_update", int(e), _ffi.ptr(x), _ffi.ptr(m8))

    def pgradient(self):
        """jastrowspin.py:457-464: the stored sums."""
        a, b, _ = self._get_state()
        return {"bcoeff": b, "acoeff": a}

    def _get_state(self):
        d = self._dev
        a, b, x = np.empty((d.W, d.natom, d.na, 2)), np.empty((d.W, d.nb, 3)), np.empty((d.W, d.N, 3))
        d.call("pqa_jastrow_get_state", _ffi.ptr(a), _ffi.ptr(b), _ffi.ptr(x))
        return a, b, x


class ThreeBodyJastrow(_DeviceFactor):
    """Electron-electron-ion Jastrow factor (protocol of ``pyqmc/wf/three_body_jastrow.py:19-655``).
    ``a_basis``/``b_basis``: lists of ``pyqmc_amd.func3d`` descriptors; parameter ``ccoeff``
    (natom, na, na, nb, 3).  The device keeps no per-electron partial sums for this factor: the one-electron sum
    P_e is re-evaluated from the stored walker coordinates, so the result does not depend on whether the driver
    moves ``configs`` before or after ``updateinternals`` (the reference's does, see tests/golden/make_golden.py)."""

    def __init__(self, mol, a_basis, b_basis, device=0, _dev=None):
        self._mol = mol
        self._nelec = int(np.sum(mol.nelec))
        if _dev is None:
            _dev = DeviceWF(mol, a3_basis=list(a_basis), b3_basis=list(b_basis), device=device)
        self._dev = _dev
        self._bind_parameters({"ccoeff": np.zeros((_dev.natom, _dev.na3, _dev.na3, _dev.nb3, 3))})
        self.dtype = float

    def recompute(self, configs):
        self.parameters.push()
        x = _ffi.f64(_xyz(self._dev, configs))
        W = x.shape[0]
        u = np.empty(W)
        self._dev.call("pqa_j3_recompute", _ffi.ptr(x), W, _ffi.ptr(u))
        self._dev.W = W
        return np.ones(W), u

    def value(self):
        u = np.empty(self._dev.W)
        self._dev.call("pqa_j3_value", _ffi.ptr(u))
        return np.ones(len(u)), u

    def _eval(self, e, epos, mask, mode):
        m, _ = _mask_args(mask, self._dev.W)
        pts, widx, aux = _points(epos, m, self._dev)
        nrow, npt = pts.shape[0], pts.shape[1]
        out = np.empty(nrow * npt) if mode == 0 else np.empty((4, nrow))
        if nrow:
            self._dev.call("pqa_j3_eval", int(e), _ffi.ptr(pts), nrow, npt, _ffi.ptr(widx), mode, _ffi.ptr(out))
        return out, nrow, npt, aux

    def testvalue(self, e, epos, mask=None):
        r, nrow, npt, aux = self._eval(e, epos, mask, 0)
        return (r.reshape(nrow, npt) if aux else r), None

    def testvalue_many(self, e, epos, mask=None):
        """three_body_jastrow.py:343-372 -> (nconf[mask], len(e))."""
        return _testvalue_many(self._dev, 4, e, epos, mask)

    def pgradient(self):
        """three_body_jastrow.py:657-719: dU/dccoeff (nconf, natom, na, na, nb, 3)."""
        d = self._dev
        out = np.empty((d.W, d.natom, d.na3, d.na3, d.nb3, 3))
        d.call("pqa_j3_pgradient", _ffi.ptr(out))
        return {"ccoeff": out}

    def gradient_value(self, e, epos):
        r, *_ = self._eval(e, epos, None, 1)
        return r[:3], r[3], None

    def gradient(self, e, epos):
        return self._eval(e, epos, None, 1)[0][:3]

    def gradient_laplacian(self, e, epos):
        r, *_ = self._eval(e, epos, None, 2)
        return r[:3], r[3]

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        _, m8 = _mask_args(mask, self._dev.W)
        x = _ffi.f64(_xyz(self._dev, epos))
        self._dev.call("pqa_j3_update", int(e), _ffi.ptr(x), _ffi.ptr(m8))


class Parameters:
    """"wf{i}{key}" view over the factors' parameter dicts (``multiplywf.py:18-68``)."""

    def __init__(self, dicts):
        self.data = {f"wf{i + 1}": d for i, d in enumerate(dicts)}
        self.wf_count = len(dicts)

    def __setitem__(self, idx, value):
        self.data[idx[:3]][idx[3:]] = value

    def __getitem__(self, idx):
        return self.data[idx[:3]][idx[3:]]

    def __iter__(self):
        return self.keys()

    def keys(self):
        for i in range(self.wf_count):
            for k in self.data[f"wf{i + 1}"].keys():
                yield f"wf{i + 1}{k}"

    def items(self):
        for k in self.keys():
            yield k, self[k]

    def values(self):
        for k in self.keys():
            yield self[k]

    def __len__(self):
        return sum(len(d) for d in self.data.values())


class MultiplyWF:
    """Product wave function (``pyqmc/wf/multiplywf.py:71-132``)."""

    def __init__(self, *wf_factors):
        self.wf_factors = list(wf_factors)
        self.parameters = Parameters([wf.parameters for wf in wf_factors])
        self.dtype = complex if any(wf.dtype == complex for wf in wf_factors) else float

    def __getstate__(self):
        return {"wf_factors": self.wf_factors}

    def __setstate__(self, d):  # the "wf{i}key" view must point at the factors' re-bound parameter dicts
        self.__init__(*d["wf_factors"])

    def __copy__(self):
        """An independent wave function on a handle of its own (``copy.copy(wf)`` in testwf.py:44,77,108): the factors
        share ONE rebuilt DeviceWF, resident walkers included."""
        import copy

        return copy.deepcopy(self)

    def fused_device(self):
        """The shared DeviceWF when every factor lives on one handle (enables the fused entry points)."""
        devs = {id(getattr(w, "_dev", None)) for w in self.wf_factors}
        d = getattr(self.wf_factors[0], "_dev", None)
        return d if len(devs) == 1 and d is not None else None

    def recompute(self, configs):
        d = self.fused_device()
        if d is not None:
            for w in self.wf_factors:
                w.parameters.push()
            return d.recompute(_xyz(d, configs))
        res = [w.recompute(configs) for w in self.wf_factors]
        return np.prod([r[0] for r in res], axis=0), np.sum([r[1] for r in res], axis=0)

    def value(self):
        res = [w.value() for w in self.wf_factors]
        return np.prod([r[0] for r in res], axis=0), np.sum([r[1] for r in res], axis=0)

    def _product_device(self, epos=None):
        """The shared handle when this is a real Slater x two-body-Jastrow product on one device (generate_wf's default): its
        protocol methods then take ONE C call each (pqa_wf_eval / pqa_wf_update) instead of one per factor."""
        f = self.wf_factors
        if len(f) != 2 or type(f[0]) is not Slater or type(f[1]) is not JastrowSpin:
            return None
        d = self.fused_device()
        if d is None or d.cplx or (epos is not None and np.ndim(epos.configs) != 2):
            return None
        return d

    def updateinternals(self, e, epos, configs, mask=None, saved_values=None):
        d = self._product_device(epos)
        if d is not None:
            sl = self.wf_factors[0]
            s = sl._spin(e)
            zero = d._zero[s]
            if zero is None:
                flag = C.c_int()
                d.call("pqa_slater_has_zero", s, C.byref(flag))
                zero = bool(flag.value)
            if not zero:
                _, m8 = _mask_args(mask, d.W)
                sv = None if saved_values is None else saved_values[0]
                use_saved = sv is not None and sv is sl._saved and sv[1] == int(e)
                d.wf_update(e, _ffi.f64(_xyz(d, epos)), m8, use_saved, s)
                sl._saved = None
                return
        saved_values = [None] * len(self.wf_factors) if saved_values is None else saved_values
        for w, sv in zip(self.wf_factors, saved_values):
            w.updateinternals(e, epos, configs, mask=mask, saved_values=sv)

    def gradient(self, e, epos):
        d = self._product_device(epos)
        if d is not None:
            r = d.wf_eval(e, _ffi.f64(_xyz(d, epos)), 1, False)
            with np.errstate(divide="ignore", invalid="ignore"):
                return r[1:4] / r[0] + r[5:8]
        return np.sum([w.gradient(e, epos) for w in self.wf_factors], axis=0)

    def testvalue(self, e, epos, mask=None):
        vals, saved = zip(*[w.testvalue(e, epos, mask=mask) for w in self.wf_factors])
        return np.prod(vals, axis=0), saved

    def pgradient(self):
        """multiplywf.py:131-132."""
        return Parameters([w.pgradient() for w in self.wf_factors])

    def testvalue_many(self, e, epos, mask=None):
        """multiplywf.py:112-114; one fused call when all factors share a device handle."""
        dev = self.fused_device()
        if dev is not None:
            bits = 0
            for w in self.wf_factors:
                bits |= {Slater: 1, JastrowSpin: 2, ThreeBodyJastrow: 4}[type(w)]
            return _testvalue_many(dev, bits, e, epos, mask)
        return np.prod([w.testvalue_many(e, epos, mask=mask) for w in self.wf_factors], axis=0)

    def gradient_value(self, e, epos):
        d = self._product_device(epos)
        if d is not None:
            r = d.wf_eval(e, _ffi.f64(_xyz(d, epos)), 1, True)
            with np.errstate(divide="ignore", invalid="ignore"):
                deriv = r[1:4] / r[0]
            deriv[~np.isfinite(deriv)] = 0.0  # (as Slater.gradient_value, slater.py:403-418)
            val = r[0].copy()
            val[~np.isfinite(val)] = 1.0
            sl = self.wf_factors[0]
            sl._saved = ("pqa-slater-saved", int(e), next(_serial))
            return deriv + r[5:8], val * r[8], (sl._saved, None)
        g, v, s = zip(*[w.gradient_value(e, epos) for w in self.wf_factors])
        return np.sum(g, axis=0), np.prod(v, axis=0), s

    def gradient_laplacian(self, e, epos):
        d = self._product_device(epos)
        if d is not None:
            r = d.wf_eval(e, _ffi.f64(_xyz(d, epos)), 2, False)
            with np.errstate(divide="ignore", invalid="ignore"):
                gs, ls = r[1:4] / r[0], r[4] / r[0]
            return gs + r[5:8], ls + r[8] + 2 * np.sum(gs * r[5:8], axis=0)
        g, l = zip(*[w.gradient_laplacian(e, epos) for w in self.wf_factors])
        cross = np.zeros(l[0].shape, dtype=self.dtype)
        for i in range(len(g)):
            for j in range(i + 1, len(g)):
                cross += np.sum(g[i] * g[j], axis=0)
        return np.sum(g, axis=0), np.sum(l, axis=0) + 2 * cross


def _ion_cusp_list(mol, ion_cusp):
    if ion_cusp is None:
        charges = mol.atom_charges()
        return [mol.atom_symbol(i) for i in range(mol.natm) if mol.atom_symbol(i) not in mol._ecp and charges[i] > 0]
    if ion_cusp is True:
        return [mol.atom_symbol(i) for i in range(mol.natm)]
    return [] if ion_cusp is False else list(ion_cusp)


def generate_jastrow(mol, ion_cusp=None, device=0, **kws):
    """Stand-alone two-body Jastrow factor with the defaults of ``wftools.generate_jastrow`` (wftools.py:99-152).
    Returns (jastrow, to_opt)."""
    ion_cusp = _ion_cusp_list(mol, ion_cusp)
    abasis, bbasis = func3d.default_jastrow_basis(mol, len(ion_cusp) > 0, **kws)
    ja = JastrowSpin(mol, abasis, bbasis, device=device)
    acoeff = np.zeros((mol.natm, len(abasis), 2))
    if ion_cusp:
        coefs = np.array(mol.atom_charges(), dtype=float)
        coefs[[mol.atom_symbol(i) not in ion_cusp for i in range(mol.natm)]] = 0.0
        acoeff[:, 0, :] = coefs[:, None]
    bcoeff = np.zeros((len(bbasis), 3))
    bcoeff[0] = [-0.25, -0.5, -0.25]
    ja.parameters["acoeff"], ja.parameters["bcoeff"] = acoeff, bcoeff
    to_opt = {"acoeff": np.ones(acoeff.shape, dtype=bool), "bcoeff": np.ones(bcoeff.shape, dtype=bool)}
    if ion_cusp:
        to_opt["acoeff"][:, 0, :] = False
    to_opt["bcoeff"][0, [0, 1, 2]] = False
    return ja, to_opt


def generate_jastrow3(mol, device=0, **kws):
    """``wftools.generate_jastrow3`` (wftools.py:155-162): default radial bases without ion cusp, zero ccoeff."""
    a3, b3 = func3d.default_jastrow_basis(mol, False, **kws)
    j3 = ThreeBodyJastrow(mol, a3, b3, device=device)
    return j3, {"ccoeff": np.ones(j3.parameters["ccoeff"].shape, dtype=bool)}


def generate_wf(mol, mf, determinants=None, jastrow_kws=None, device=0, tol=None, jastrow3=False, jastrow3_kws=None,
                eval_gto_precision=None, image_rule="reference"):
    """Slater x two-body-Jastrow product on ONE device handle — the counterpart of
    ``pyqmc.wftools.generate_wf`` (wftools.py:195-241) with the default Jastrow of
    ``generate_jastrow`` (:99-152: e-e cusp fixed at -1/4, -1/2, -1/4; ion cusp only for
    all-electron ions)."""
    kws = dict(jastrow_kws or {})
    ion_cusp = kws.pop("ion_cusp", None)
    if ion_cusp is None:
        charges = mol.atom_charges()
        ion_cusp = [mol.atom_symbol(i) for i in range(mol.natm) if mol.atom_symbol(i) not in mol._ecp and charges[i] > 0]
    elif ion_cusp is True:
        ion_cusp = [mol.atom_symbol(i) for i in range(mol.natm)]
    elif ion_cusp is False:
        ion_cusp = []
    abasis, bbasis = func3d.default_jastrow_basis(mol, len(ion_cusp) > 0, **kws)
    mol, mo_coeff, determinants, twist_k, fold = orbital_inputs(mol, mf, determinants, with_fold=True)
    a3 = b3 = None
    if jastrow3:  # wftools.generate_jastrow3 (:155-162): default basis without ion cusp
        a3, b3 = func3d.default_jastrow_basis(mol, False, **dict(jastrow3_kws or {}))
    dev = DeviceWF(mol, mo_coeff=mo_coeff, determinants=determinants, a_basis=abasis, b_basis=bbasis, device=device,
                   tol=-1 if tol is None else tol, a3_basis=a3, b3_basis=b3, eval_gto_precision=eval_gto_precision,
                   image_rule=image_rule, twist_k=twist_k)
    sl = Slater(mol, mf, _dev=dev, _fold=fold)
    ja = JastrowSpin(mol, abasis, bbasis, _dev=dev)
    acoeff = np.zeros((mol.natm, len(abasis), 2))
    if ion_cusp:
        coefs = np.array(mol.atom_charges(), dtype=float)
        coefs[[mol.atom_symbol(i) not in ion_cusp for i in range(mol.natm)]] = 0.0
        acoeff[:, 0, :] = coefs[:, None]
    bcoeff = np.zeros((len(bbasis), 3))
    bcoeff[0] = [-0.25, -0.5, -0.25]
    ja.parameters["acoeff"] = acoeff
    ja.parameters["bcoeff"] = bcoeff
    if jastrow3:
        return MultiplyWF(sl, ja, ThreeBodyJastrow(mol, a3, b3, _dev=dev))
    return MultiplyWF(sl, ja)
