"""Host logic of the DMC driver (pyqmc_amd/dmc.py) and of the distributed branching, on CPU.

The driver only talks to the wave-function protocol and the accumulator interface, so here the ORACLE
objects stand in for the GPU engine; the replayed random draws are the reference's own
(tests/golden/g12_dmc.npz), so the protocol-route harness (tests/helpers.protocol_dmc_propagate: the reference's control flow
over the protocol) must reproduce the reference's dmc_propagate exactly; rundmc's block loop, restart files and the
distributed comb are the product's (pyqmc_amd.dmc / dist), driven here with the harness as propagator."""

import os
import socket

import numpy as np

import helpers
from helpers import golden, relerr
from pyqmc_amd import dist as pdist
from pyqmc_amd import dmc, systems
from pyqmc_amd.configs import OpenConfigs


class OracleAccumulator:
    """EnergyAccumulator interface over the oracle (test double for the device accumulator)."""

    def __init__(self, mol, threshold=10.0, ewald_kws=None):
        self.mol, self.threshold, self.ewald_kws = mol, threshold, ewald_kws

    class _Dev:
        def __init__(self, necp):
            self.necp = necp

    def _device(self, wf):
        from oracle import energy as oenergy

        return self._Dev(len(oenergy.ecp_atoms(self.mol)))

    def __call__(self, configs, wf, rot=None, unif=None):
        from oracle import energy as oenergy

        if rot is None:  # no replay tape: draw like the reference does, from numpy's global generator
            W, N = configs.configs.shape[:2]
            necp, rng = self._device(wf).necp, helpers.NumpyRNG()
            unif = np.array([[rng.random(W) for _ in range(necp)] for _ in range(N)])
            rot = np.array([[rng.rot() for _ in range(necp)] for _ in range(N)])
        return oenergy.energy(self.mol, configs, wf, self.threshold, rot, unif, ewald_kws=self.ewald_kws)

    def avg(self, configs, wf):  # accumulators.py:77-84
        return {k: np.mean(v, axis=0) for k, v in self(configs, wf).items()}

    def has_nonlocal_moves(self):
        return bool(self.mol._ecp)

    def nonlocal_tmoves(self, configs, wf, e, tau, rot=None, unif=None):
        from oracle import dmc as odmc

        class T:
            def __init__(s):
                s.r, s.u = iter(rot), iter(unif)

            def rot(s):
                return next(s.r)

            def random(s, W):
                return next(s.u)

        if rot is None:
            W, necp, rng = configs.configs.shape[0], self._device(wf).necp, helpers.NumpyRNG()
            unif, rot = [rng.random(W) for _ in range(necp)], [rng.rot() for _ in range(necp)]
        ratio, weight, pos = odmc.compute_tmoves(self.mol, configs, wf, e, self.threshold, tau, T())
        return {"ratio": ratio, "weight": weight, "configs": configs.make_irreducible(e, pos)}


def test_driver_reproduces_reference_dmc_propagate():
    g = golden("g12_dmc")
    mol = systems.water()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    accepts = []
    orig = wf.updateinternals
    wf.updateinternals = lambda e, ep, c, mask=None, saved_values=None: (accepts.append(np.asarray(mask).copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1]
    df, configs, weights = helpers.protocol_dmc_propagate(wf, OpenConfigs(g["start"].copy()), g["weights0"].copy(), float(tstep), float(branchcut),
                                             float(e_trial), float(e_est), nsteps=int(nsteps),
                                             accumulators={"energy": OracleAccumulator(mol)}, rng=helpers.ReplayTape(g))
    assert np.array_equal(np.asarray(accepts), g["accepts"])
    assert relerr(configs.configs, g["final"]) < 1e-10 and relerr(weights, g["weights"]) < 1e-9
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert relerr(df[k], g["df_" + k]) < 1e-9, k


def test_limdrift_and_compute_S_match_oracle():
    from oracle import dmc as odmc

    rng = np.random.default_rng(0)
    gvec = rng.standard_normal((50, 3)) * np.logspace(-6, 2, 50)[:, None]
    assert np.allclose(helpers.limdrift(gvec, 0.02), odmc.limdrift(gvec, 0.02), rtol=1e-14, atol=0)
    v2, eloc = rng.random(20) * 50, rng.standard_normal(20) * 5
    assert np.allclose(helpers.compute_S(-1.0, -1.2, 2.0, v2, 0.02, eloc, 8), odmc.compute_S(-1.0, -1.2, 2.0, v2, 0.02, eloc.copy(), 8), rtol=1e-14)


def test_branch_matches_reference_golden():
    g = golden("g12_dmc")
    cfg = OpenConfigs(g["branch_configs"].copy())
    cfg, w, info = dmc.branch(cfg, g["branch_weights"].copy(), float(g["branch_u"]))
    assert np.array_equal(cfg.configs, g["branch_newconfigs"]) and relerr(w, g["branch_newweights"]) < 1e-14
    assert [info["max branches"], info["Number of walkers killed"]] == g["branch_info"].tolist()


def _periodic_branch_input(g):
    """The golden ensemble placed in a big periodic box, with distinguishable wrap counters."""
    from pyqmc_amd.configs import PeriodicConfigs

    x = g["branch_configs"]
    lat = np.eye(3) * 1000.0
    wrap = np.arange(x.size, dtype=float).reshape(x.shape) % 7 - 3
    return PeriodicConfigs(x + 500.0, lat, wrap=wrap.copy()), wrap


def _branch_worker(rank, world, port, q, periodic=False):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = golden("g12_dmc")
        lo, hi = pdist.shard_bounds(len(g["branch_weights"]), world)[rank]
        if periodic:
            cfg = _periodic_branch_input(g)[0].mask(slice(lo, hi))
        else:
            cfg = OpenConfigs(g["branch_configs"][lo:hi].copy())
        cfg, w, info, wstd = pdist.branch_distributed(cfg, g["branch_weights"][lo:hi].copy(), base_u=float(g["branch_u"]))
        q.put((rank, cfg.configs, w, info, wstd, getattr(cfg, "wrap", None)))
    finally:
        dist.destroy_process_group()


def test_distributed_branch_two_ranks_gloo():
    """Sharded stochastic comb == the reference's gather -> branch -> re-split (dmc.py:286-287,342-376,566)."""
    import torch.multiprocessing as mp

    g = golden("g12_dmc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_branch_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # together the ranks hold exactly the reference's resampled ensemble, bit for bit, as a multiset (walkers are exchangeable
    # and carry equal weights after the comb; slots are filled in source order so that few walkers change ranks)
    assert np.array_equal(_sorted_rows(np.concatenate([r[1] for r in res])), _sorted_rows(g["branch_newconfigs"]))
    assert relerr(np.concatenate([r[2] for r in res]), g["branch_newweights"]) < 1e-14
    for r in res:
        assert [r[3]["max branches"], r[3]["Number of walkers killed"]] == g["branch_info"].tolist()
        assert abs(r[4] - np.std(g["branch_weights"])) < 1e-14


def _sorted_rows(x):
    rows = np.asarray(x).reshape(len(x), -1)
    return rows[np.lexsort(rows.T[::-1])]


def test_distributed_branch_four_ranks_moves_only_reassigned_walkers():
    """weights all-gathered, identical comb everywhere, and ONLY the walkers whose new owner differs from the old
    one cross ranks.  World size 4 over gloo: every rank ends with exactly its slice of the single-process comb (bit for bit),
    the number of walkers that crossed equals the plan's count and the bytes sent are those walkers' coordinates only."""
    import torch.multiprocessing as mp

    g = golden("g12_dmc")
    W = len(g["branch_weights"])
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_branch_worker, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    bounds = pdist.shard_bounds(W, 4)
    counts = [hi - lo for lo, hi in bounds]
    newinds = np.sort(dmc.comb_indices(g["branch_weights"], float(g["branch_u"]))[0])  # slots filled in source order
    assert np.array_equal(_sorted_rows(np.concatenate([r[1] for r in res])), _sorted_rows(g["branch_newconfigs"]))  # = the reference's ensemble
    owner_new = np.repeat(np.arange(4), counts)
    owner_old = np.searchsorted(np.cumsum(counts), newinds, side="right")
    moved = int(np.sum(owner_new != owner_old))
    assert 0 < moved < W // 2  # the fixture's weights do send some walkers across, and keep most at home
    sent = 0
    for r, (lo, hi) in zip(res, bounds):
        assert np.array_equal(_sorted_rows(r[1]), _sorted_rows(g["branch_configs"][newinds[lo:hi]]))
        assert r[3]["walkers moved"] == moved and len(r[1]) == hi - lo
        keep, recv, send = pdist.exchange_plan(newinds, counts, r[0])
        assert r[3]["bytes exchanged"] == 8 * g["branch_configs"][0].size * sum(len(v) for v in send.values())
        assert len(keep) + sum(recv.values()) == hi - lo
        sent += r[3]["bytes exchanged"]
    assert sent == moved * 8 * g["branch_configs"][0].size  # nothing but the re-assigned walkers travelled


def test_distributed_branch_periodic_walkers_keep_their_wrap_counters():
    """PeriodicConfigs through the sharded comb: coordinates AND wrap counters of every surviving walker arrive
    together (the reference resamples both arrays with the same indices, coord.py:191-198)."""
    import torch.multiprocessing as mp

    g = golden("g12_dmc")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_branch_worker, args=(r, 2, port, q, True)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg0, wrap0 = _periodic_branch_input(g)
    newinds, _ = dmc.comb_indices(g["branch_weights"], float(g["branch_u"]))
    # (mask() re-folds through the constructor like the reference, coord.py:159-162: equal to the last bit or two)
    n = len(newinds)
    both = np.concatenate([np.concatenate([r[1] for r in res]).reshape(n, -1), np.concatenate([r[5] for r in res]).reshape(n, -1)], axis=1)
    ref = np.concatenate([cfg0.configs[newinds].reshape(n, -1), cfg0.wrap[newinds].reshape(n, -1)], axis=1)  # a walker's wrap counters stay with it
    assert np.allclose(_sorted_rows(np.round(both, 8)), _sorted_rows(np.round(ref, 8)), rtol=0, atol=1e-8)


def test_driver_reproduces_reference_periodic_dmc_propagate():
    """dmc_propagate on PeriodicConfigs (diamond, Ewald energies, T-moves through make_irreducible): the driver over the
    oracle reproduces the reference's run, wrap counters included (tests/golden/g17_pbc_dmc.npz)."""
    from pyqmc_amd.configs import PeriodicConfigs

    g = golden("g17_pbc_dmc")
    sup, wf = helpers.oracle_pbc_wf("gamma")
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    accepts = []
    orig = wf.updateinternals
    wf.updateinternals = lambda e, ep, c, mask=None, saved_values=None: (accepts.append(np.asarray(mask).copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = helpers.protocol_dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                         nsteps=int(nsteps), accumulators={"energy": OracleAccumulator(sup)}, rng=helpers.ReplayTape(g))
    assert np.array_equal(np.asarray(accepts), g["accepts"])
    assert relerr(cfg.configs, g["final"]) < 1e-10 and np.array_equal(cfg.wrap, g["final_wrap"]) and relerr(weights, g["weights"]) < 1e-9
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert relerr(df[k], g["df_" + k]) < 1e-9, k


def test_protocol_route_reproduces_reference_complex_dmc_propagate():
    """Complex wave function (3x1x1 diamond supercell, complex Bloch coefficients) through dmc_propagate with ECP T-moves: the
    protocol-route harness over the oracle reproduces the reference run of golden g30 — accept decisions of both phases, walkers,
    wrap counters, weights, complex block averages.  The T-move amplitudes use Re[Psi(R')/Psi(R)] (oracle/dmc.py:select_tmoves,
    make_golden.py:g_pbc_complex_dmc say why); no node constraint in the drift-diffusion (dmc.py:64-66 is real-only)."""
    from pyqmc_amd.configs import PeriodicConfigs
    from test_pbc_cpu import _oracle_complex_wf

    g = golden("g30_pbc_complex_dmc")
    sup, wf = _oracle_complex_wf(golden("g19_pbc_complex"))
    assert wf.dtype == complex
    tstep, branchcut, e_trial, e_est, nsteps = g["params"]
    accepts = []
    orig = wf.updateinternals
    wf.updateinternals = lambda e, ep, c, mask=None, saved_values=None: (accepts.append(np.asarray(mask).copy()), orig(e, ep, c, mask=mask, saved_values=saved_values))[1]
    cfg = PeriodicConfigs(g["start"].copy(), sup.lattice_vectors(), wrap=g["start_wrap"].copy())
    df, cfg, weights = helpers.protocol_dmc_propagate(wf, cfg, g["weights0"].copy(), float(tstep), float(branchcut), float(e_trial), float(e_est),
                                                      nsteps=int(nsteps), accumulators={"energy": OracleAccumulator(sup, ewald_kws={"ewald_gmax": 10})},
                                                      rng=helpers.ReplayTape(g))
    assert np.array_equal(np.asarray(accepts), g["accepts"]) and g["accepts"][: 2 * int(sum(sup.nelec))].sum() > 0
    assert relerr(cfg.configs, g["final"]) < 1e-10 and np.array_equal(cfg.wrap, g["final_wrap"]) and relerr(weights, g["weights"]) < 1e-9
    assert set(df.keys()) == set(g["df_keys"].tolist())
    for k in df:
        assert relerr(df[k], g["df_" + k]) < 1e-9, k
    assert abs(np.imag(df["energytotal"])) > 1e-3  # a genuinely complex local energy went through the averages


def _small_dmc(path, nblocks, W=6, seed=3, distributed=False, **kw):
    """rundmc over the ORACLE wave function (protocol route) on a handful of H2O walkers, block file at `path`."""
    import pyqmc_amd as pa

    mol = systems.water()
    wf = helpers.oracle_wf(mol, systems.random_mf(mol))
    np.random.seed(seed)
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(seed))
    return dmc.rundmc(wf, cfg, tstep=0.05, nblocks=nblocks, nsteps_per_block=1, vmc_warmup=1, distributed=distributed,
                      accumulators={"energy": OracleAccumulator(mol)}, hdf_file=path, propagate=helpers.protocol_dmc_propagate,
                      vmc_worker=helpers.protocol_vmc_worker, **kw)


def test_rundmc_continues_an_existing_block_file(tmp_path):
    """dmc.py:466-500: a second rundmc on the same hdf_file restarts from the stored walkers and weights, takes e_trial, e_est
    and esigma from the file's last block, numbers its blocks from block[-1] + 1 and keeps estimating the energy from the
    whole history; continue_from + an existing hdf_file is refused."""
    import pytest

    from pyqmc_amd import blockfile

    path = str(tmp_path / "dmc.hdf5")
    df1, cfg1, w1 = _small_dmc(path, 2)
    store = blockfile.BlockFile(path)
    assert store.datasets()["block"].tolist() == [0, 1]
    df2, cfg2, w2 = _small_dmc(path, 4, seed=99)  # (other start walkers: they must be replaced by the file's)
    d = store.datasets()
    assert df2["block"].tolist() == [2, 3] and d["block"].tolist() == [0, 1, 2, 3]
    # the first continued block ran with the parameters the file's last block recorded (dmc.py:489-491) ...
    assert df2["e_trial"][0] == df1["e_trial"][-1] and df2["e_est"][0] == df1["e_est"][-1] and df2["esigma"][0] == df1["esigma"][-1]
    # ... and the estimate after it averages the whole record (estimate_energy reads the file, dmc.py:594-603)
    en, wt = d["energytotal"][:3], d["weight"][:3]
    assert abs(df2["e_est"][1] - np.average(en[0:], weights=wt[0:]).real) < 1e-12
    assert np.array_equal(store._state()["configs"], cfg2.configs) and np.array_equal(store._state()["weights"], w2)
    assert _small_dmc(path, 4)[0] == {}  # nothing left to run
    with pytest.raises(RuntimeError, match="already exists"):
        _small_dmc(path, 6, continue_from=path)
    # continue_from into a NEW file: offsets and walkers from the old one, record in the new one only
    path2 = str(tmp_path / "dmc2.hdf5")
    df3, _, _ = _small_dmc(path2, 5, continue_from=path)
    assert df3["block"].tolist() == [4] and blockfile.BlockFile(path2).datasets()["block"].tolist() == [4]


def _rundmc_worker(rank, world, port, path, q):
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        df, cfg, w = _small_dmc(path, 2, W=4, seed=10 + rank, distributed=True)
        df2, cfg2, w2 = _small_dmc(path, 3, W=4, seed=50 + rank, distributed=True)
        q.put((rank, df["block"].tolist(), df2["block"].tolist(), cfg2.configs, w2, df2["energytotal"].tolist()))
    finally:
        dist.destroy_process_group()


def test_sharded_rundmc_writes_one_block_file_per_rank():
    """N ranks must not append to one file.  Rank 0 owns hdf_file, rank r hdf_file.rank<r>; each holds the common
    (all-reduced) block record and its OWN shard's walkers and weights, and each rank continues from its own file."""
    import tempfile

    import torch.multiprocessing as mp

    from pyqmc_amd import blockfile

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "dmc.hdf5")
        procs = [ctx.Process(target=_rundmc_worker, args=(r, 2, port, path, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
        stores = [blockfile.BlockFile(path), blockfile.BlockFile(path + ".rank1")]
        for r, st in zip(res, stores):
            assert st.exists() and r[1] == [0, 1] and r[2] == [2]
            d = st.datasets()
            assert d["block"].tolist() == [0, 1, 2]
            assert np.array_equal(st._state()["configs"], r[3]) and np.array_equal(st._state()["weights"], r[4])
        # the block record is the all-reduced one: identical in both files; the shards are not
        assert np.array_equal(stores[0].datasets()["energytotal"], stores[1].datasets()["energytotal"])
        assert not np.array_equal(res[0][3], res[1][3])


def _fake_propagate(rank):
    def propagate(wf, configs, weights, tstep, branchcut, e_trial, e_est, nsteps=1, accumulators=None, ekey=None):
        blk = {"energytotal": complex(-17.0 - rank, 0.25 + 0.5 * rank), "energyke": 3.0 + rank, "obdmvalue": (rank + 1.0) * np.arange(6.0).reshape(2, 3),
               "weight": 1.0 + 0.5 * rank, "acceptance": 0.5, "tmove_acceptance": 0.0}
        return blk, configs, weights
    return propagate


def _complex_block_worker(rank, world, port, q):
    import torch.distributed as dist

    import pyqmc_amd as pa

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        means, n = pdist.allreduce_block(np.array([1.0 + 2.0j * (rank + 1), 3.0 * (rank + 1)]), 2 + rank)
        mol = systems.water()
        wf = helpers.oracle_wf(mol, systems.random_mf(mol))
        np.random.seed(rank)
        cfg = pa.initial_guess(mol, 4 + rank, rng=np.random.default_rng(rank))
        df, _, _ = dmc.rundmc(wf, cfg, tstep=0.05, nblocks=1, nsteps_per_block=1, vmc_warmup=1, distributed=True,
                              accumulators={"energy": OracleAccumulator(mol)}, propagate=_fake_propagate(rank), vmc_worker=helpers.protocol_vmc_worker)
        q.put((rank, means, n, {k: np.asarray(v) for k, v in df.items()}))
    finally:
        dist.destroy_process_group()


def test_sharded_rundmc_recombines_complex_and_array_valued_blocks():
    """a complex (twisted) wave function's block carries complex <acc>ecp / <acc>total, and host accumulators
    (density matrices) carry arrays: the sharded recombination reduces real and imaginary parts (RCCL has no complex type) and
    keeps shapes — weight-averaged over the ranks' shards exactly as dmc.py:238-304 combines worker blocks."""
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_complex_block_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, means, n, df in res:
        assert n == 5 and np.iscomplexobj(means)
        assert np.allclose(means, [(2.0 + 6.0j) / 5, 9.0 / 5])
        # ranks hold 4 and 5 walkers with block weights 1.0 and 1.5: weight of a rank in the block average = weight * walkers
        wr = np.array([1.0 * 4, 1.5 * 5])
        wr = wr / wr.sum()
        assert np.allclose(df["energytotal"][0], wr[0] * complex(-17.0, 0.25) + wr[1] * complex(-18.0, 0.75))
        assert df["energytotal"].dtype.kind == "c" and df["energyke"].dtype.kind == "f"
        assert np.allclose(df["energyke"][0], wr[0] * 3.0 + wr[1] * 4.0)
        assert df["obdmvalue"].shape == (1, 2, 3) and np.allclose(df["obdmvalue"][0], (wr[0] + 2.0 * wr[1]) * np.arange(6.0).reshape(2, 3))
        assert np.allclose(df["weight"][0], (1.0 * 4 + 1.5 * 5) / 9)
