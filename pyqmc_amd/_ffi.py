"""ctypes binding of the C ABI declared in ``include/pyqmc_amd.h``.

The HIP library is the product path: if it is missing or cannot be loaded this
module raises — there is no CPU fallback anywhere in ``pyqmc_amd``.
"""

import ctypes as C
import os
import re

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PQA_LIB") or os.path.join(_HERE, "lib", "libpyqmc_amd.so")  # PQA_LIB: A/B of two builds (tools/)
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "pyqmc_amd.h")

c_double_p = C.POINTER(C.c_double)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class DmcTapes(C.Structure):
    """pqa_dmc_tapes_t (include/pyqmc_amd.h)."""

    _fields_ = [(n, C.c_void_p) for n in ("gauss", "unif", "tm_rot", "tm_unif", "tm_u1", "tm_u2", "ecp_rot", "ecp_unif")]


class SystemStruct(C.Structure):
    """Mirror of ``pqa_system_t``."""

    _fields_ = [
        ("natom", C.c_int32), ("nelec_up", C.c_int32), ("nelec_dn", C.c_int32),
        ("atom_xyz", c_double_p), ("atom_charge", c_double_p),
        ("nshell", C.c_int32), ("nprim", C.c_int32), ("nao", C.c_int32),
        ("shell_atom", c_int32_p), ("shell_l", c_int32_p), ("shell_prim_off", c_int32_p), ("shell_ao_off", c_int32_p),
        ("prim_exp", c_double_p), ("prim_coef", c_double_p),
        ("nmo_up", C.c_int32), ("nmo_dn", C.c_int32), ("mo_up", c_double_p), ("mo_dn", c_double_p),
        ("ndet", C.c_int32), ("ndet_up", C.c_int32), ("ndet_dn", C.c_int32),
        ("det_coeff", c_double_p), ("det_occ_up", c_int32_p), ("det_occ_dn", c_int32_p), ("det_map", c_int32_p),
        ("na", C.c_int32), ("nb", C.c_int32),
        ("a_kind", c_int32_p), ("a_param", c_double_p), ("b_kind", c_int32_p), ("b_param", c_double_p),
        ("rcut_a", C.c_double), ("rcut_b", C.c_double), ("acoeff", c_double_p), ("bcoeff", c_double_p),
        ("necp", C.c_int32), ("ecp_atom", c_int32_p), ("ecp_chan_off", c_int32_p), ("ecp_term_off", c_int32_p),
        ("ecp_term_n", c_int32_p), ("ecp_term_exp", c_double_p), ("ecp_term_coef", c_double_p),
        ("has_slater", C.c_int32),
        ("na3", C.c_int32), ("nb3", C.c_int32),
        ("a3_kind", c_int32_p), ("a3_param", c_double_p), ("b3_kind", c_int32_p), ("b3_param", c_double_p),
        ("rcut_a3", C.c_double), ("rcut_b3", C.c_double), ("ccoeff", c_double_p),
        ("pbc", C.c_int32), ("lattice", C.c_double * 9),
        ("nL", C.c_int32), ("Ls", c_double_p), ("num_Ls", c_int32_p), ("atom_cut", c_double_p), ("shell_cut", c_double_p),
        ("lattice_prim", C.c_double * 9), ("img_n", c_int32_p), ("atom_n", c_int32_p), ("member", C.POINTER(C.c_uint8)),
        ("member_class", c_int32_p), ("member_M", C.c_int32), ("n_member_class", C.c_int32),
        ("complex_orbitals", C.c_int32), ("twisted", C.c_int32), ("twist_k", C.c_double * 3),
    ]


_H = C.c_void_p
_PROTOTYPES = {
    "pqa_create": (C.c_int, [C.POINTER(SystemStruct), C.c_int, C.POINTER(_H)]),
    "pqa_destroy": (None, [_H]),
    "pqa_last_error": (C.c_char_p, [_H]),
    "pqa_device_count": (C.c_int, []),
    "pqa_set_param": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_int64]),
    "pqa_get_param": (C.c_int, [_H, C.c_char_p, C.c_void_p, C.c_int64]),
    "pqa_slater_pgradient": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pqa_j3_pgradient": (C.c_int, [_H, C.c_void_p]),
    "pqa_testvalue_many": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_void_p]),
    "pqa_set_ewald": (C.c_int, [_H, C.c_double, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                C.c_void_p, C.c_void_p]),
    "pqa_get_wrap": (C.c_int, [_H, C.c_void_p]),
    "pqa_eval_ao": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "pqa_eval_mo": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_int, C.c_void_p]),
    "pqa_slater_recompute": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pqa_slater_value": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "pqa_slater_eval": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pqa_slater_update": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "pqa_slater_has_zero": (C.c_int, [_H, C.c_int, C.POINTER(C.c_int)]),
    "pqa_slater_get_state": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "pqa_jastrow_recompute": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p]),
    "pqa_jastrow_value": (C.c_int, [_H, C.c_void_p]),
    "pqa_jastrow_eval": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pqa_jastrow_update": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "pqa_jastrow_get_state": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pqa_j3_recompute": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p]),
    "pqa_j3_value": (C.c_int, [_H, C.c_void_p]),
    "pqa_j3_eval": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "pqa_j3_update": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p]),
    "pqa_wf_recompute": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pqa_wf_value": (C.c_int, [_H, C.c_void_p, C.c_void_p]),
    "pqa_get_configs": (C.c_int, [_H, C.c_void_p]),
    "pqa_energy": (C.c_int, [_H, C.c_double, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "pqa_tmove_npoints": (C.c_int, [_H]),
    "pqa_tmoves": (C.c_int, [_H, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pqa_vmc_sweeps": (C.c_int, [_H, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pqa_resample": (C.c_int, [_H, C.c_void_p]),
    "pqa_get_walkers": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p]),
    "pqa_branch_exchange": (C.c_int, [_H, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64]),
    "pqa_dmc_steps": (C.c_int, [_H, C.c_double, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p,
                                C.c_uint64, C.c_void_p, C.c_void_p]),
    "pqa_dm_walk": (C.c_int, [_H, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int,
                              C.c_void_p, C.c_void_p]),
    "pqa_dm_points": (C.c_int, [_H, C.c_int, C.c_int, C.c_void_p, C.c_int64]),
    "pqa_obdm_accumulate": (C.c_int, [_H, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "pqa_tbdm_accumulate": (C.c_int, [_H, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                      C.c_int, C.c_int]),
    "pqa_dm_fetch": (C.c_int, [_H, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p]),
    "pqa_gram": (C.c_int, [_H, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pqa_philox_tapes": (C.c_int, [_H, C.c_uint64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p]),
    "pqa_philox_dmc_tapes": (C.c_int, [_H, C.c_uint64, C.c_int, C.c_int64, C.c_void_p]),
    "pqa_timer_start": (C.c_int, [_H]),
    "pqa_timer_stop": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "pqa_sync": (C.c_int, [_H]),
    "pqa_profile_enable": (C.c_int, [_H, C.c_int]),
    "pqa_profile_query": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "pqa_profile_query_commit": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "pqa_profile_query_part": (C.c_int, [_H, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "pqa_last_ecp_points": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "pqa_set_ecp_naip": (C.c_int, [_H, C.c_int32]),
    "pqa_dmc_continue": (C.c_int, [_H, C.c_int]),
    "pqa_dmc_can_continue": (C.c_int, [_H]),
    "pqa_wf_eval": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "pqa_wf_update": (C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]),
    "pqa_set_ecp_batched": (C.c_int, [_H, C.c_int32, C.c_void_p, C.c_int32, C.c_int32]),
    "pqa_ecp_batched_nselected": (C.c_int, [_H]),
    "pqa_ecp_batched_moves": (C.c_int, [_H, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p]),
}


def header_symbols():
    """Every function name declared in include/pyqmc_amd.h."""
    text = open(HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pqa_[a-z0-9_]+)\s*\(", text)))


_lib = None


def lib():
    """Load (once) the HIP shared library; fail loudly if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension is the only compute path of pyqmc_amd. "
                "Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
            )
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


class PqaError(RuntimeError):
    pass


def check(handle, rc):
    if rc != 0:
        msg = lib().pqa_last_error(handle)
        raise PqaError(f"pyqmc_amd error {rc}: {msg.decode() if msg else '?'}")


def ptr(a):
    """void* of a C-contiguous numpy array (None -> NULL)."""
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)
