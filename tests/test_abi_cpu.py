"""CPU-side checks: the C-ABI library builds, loads and exports every symbol declared in
include/pyqmc_amd.h; host-side table construction matches the oracle's; the product path
fails loudly without a GPU (no fallback)."""

import ctypes

import numpy as np
import pytest

import helpers  # noqa: F401
from pyqmc_amd import _ffi, systems, tables


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge

    ge.build()
    return _ffi.lib()


def test_library_exports_header_symbols(lib):
    names = _ffi.header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), n
    assert set(names) == set(_ffi._PROTOTYPES), set(names) ^ set(_ffi._PROTOTYPES)


def test_struct_layout_matches_header():
    """Field order of the ctypes mirror == field order of pqa_system_t in the header."""
    import re

    text = open(_ffi.HEADER_PATH).read()
    body = re.search(r"typedef struct \{(.*?)\} pqa_system_t;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
        fields += [re.sub(r"\[\d+\]$", "", n.strip().lstrip("*")) for n in names.split(",")]
    assert fields == [f[0] for f in _ffi.SystemStruct._fields_]


def test_no_gpu_means_loud_failure(lib):
    if lib.pqa_device_count() > 0:
        pytest.skip("a GPU is visible")
    import pyqmc_amd as pa

    mol = systems.water()
    with pytest.raises(_ffi.PqaError):
        pa.generate_wf(mol, systems.random_mf(mol))


def test_basis_tables_match_oracle():
    from oracle import gto

    for mol in (systems.water(), systems.water_cluster(), systems.carbon_dimer(), systems.helium()):
        bt = tables.basis_tables(mol)
        tab = gto.AOTable(mol)
        assert bt["nao"] == tab.nao == mol.nao()
        for k, (ia, l, exps, coefs, off) in enumerate(tab.shells):
            sl = slice(bt["shell_prim_off"][k], bt["shell_prim_off"][k + 1])
            assert (bt["shell_atom"][k], bt["shell_l"][k], bt["shell_ao_off"][k]) == (ia, l, off)
            assert np.allclose(bt["prim_exp"][sl], exps, rtol=0, atol=0)
            assert np.allclose(bt["prim_coef"][sl], coefs, rtol=1e-14)
    assert systems.water_cluster().nao() == 184 and sum(systems.water_cluster().nelec) == 64


def test_determinant_packing_matches_reference_fixture():
    g = helpers.golden("g8_protocol_h2o_multidet")
    import ast

    dets = ast.literal_eval(str(g["det_json"]))
    coef, up, dn, dmap = tables.pack_determinants((4, 4), dets)
    assert np.array_equal(up, g["det_occup_up"]) and np.array_equal(dn, g["det_occup_dn"])
    assert np.array_equal(dmap, g["det_map"]) and np.allclose(coef, g["det_coeff"])


def test_ecp_tables():
    mol = systems.water()
    et = tables.ecp_tables(mol)
    assert list(et["ecp_atom"]) == [0, 1, 2] and list(et["ecp_chan_off"]) == [0, 2, 4, 6]
    # oxygen: s channel first (1 term r^0), local last (3 terms r^-1, r^0, r^1)
    assert list(et["ecp_term_n"][:4]) == [0, -1, 0, 1]
    assert np.isclose(et["ecp_term_coef"][0], 85.86406) and np.isclose(et["ecp_term_coef"][1], 6.0)


def test_default_jastrow_basis_and_configs_container():
    import pyqmc_amd as pa

    ab, bb = pa.default_jastrow_basis(systems.water())
    assert [b.kind for b in bb] == [1, 0, 0, 0] and [b.kind for b in ab] == [0, 0, 0, 0] and bb[0].rcut == 7.5
    c = pa.OpenConfigs(np.arange(24.0).reshape(2, 4, 3))
    e = c.make_irreducible(1, c.configs[:, 1] + 1.0)
    c.move(1, e, np.array([True, False]))
    assert np.array_equal(c.configs[0, 1], [4.0, 5.0, 6.0]) and np.array_equal(c.configs[1, 1], [15.0, 16.0, 17.0])
    d, ij = c.dist.dist_matrix(c.configs)
    assert d.shape == (2, 6, 3) and ij[0] == (0, 1) and np.array_equal(d[:, 0], c.configs[:, 0] - c.configs[:, 1])
    assert [x.configs.shape[0] for x in pa.OpenConfigs(np.zeros((5, 2, 3))).split(2)] == [3, 2]


def test_committed_bench_line_follows_the_contract():
    """profiles/r01_bench.json is the line `python bench.py` printed on MI355X for this tree: the keys of the driver's
    contract, the roofline object of the dominant compute kernel and the CPU baseline object must be there."""
    import json
    import os

    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r01_bench.json")
    d = json.load(open(path))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "walker-steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f64"
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["global_walkers"] / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac"] < 1
    h = d["roofline_hbm"]
    assert h["bound"] == "hbm" and h["peak"] == 8000.0 and 0 < h["frac"] < 1
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and c["value"] > 0 and "sample" in c
