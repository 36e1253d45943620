"""Headline benchmark: walker-steps/sec of the VMC hot path on the 64-electron (H2O)8
Slater-Jastrow system (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps 8 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one full single-electron-move sweep of every walker (mc.py:115-137) plus one
EnergyAccumulator evaluation (mc.py:142-148), all device-resident (pqa_vmc_sweeps).
Walkers shard across ranks with no data-path collective; the only exchange is the
per-block all-reduce of the energy sums (RCCL), which is inside the timed region.
Prints ONE JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# fp64 peaks of MI355X (AMD datasheet; MI355X_MICROARCH.md lists no fp64 row): vector = matrix = 78.6 TFLOP/s
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_MFMA_PEAK_TFLOPS = 78.6
# what the part sustains (so `frac` can be read both ways): plain coalesced streaming reads 5.8-6.5 TB/s on these boxes
# (tools/scratch/bw_probe.hip; the guide says ~6.3), v_mfma_f64_16x16x4_f64 alone 72 TFLOP/s at 8 waves/SIMD (tools/ubench.hip)
HBM_ACHIEVABLE_GBS = 6300.0
FP64_MFMA_ACHIEVABLE_TFLOPS = 72.0


def _newest(pattern):
    """profiles/rNN_<pattern> of the latest round that has it (tools/refresh_evidence.sh writes them)."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return found[-1] if found else os.path.join(ROOT, "profiles", "r05_" + pattern)


PMC_SUMMARY = _newest("pmc_summary.json")  # tools/refresh_evidence.sh -> tools/pmc_summary.py
KERNEL_STATS = _newest("bench_kernel_stats.csv")  # rocprofv3 --kernel-trace --stats of this command (tools/prof_stats.py)


def pmc_kernel(key, walkers):
    """Counter-measured HBM bytes per launch of a kernel class, scaled to this run's walker count: rocprofv3 --pmc FETCH_SIZE
    and --pmc WRITE_SIZE (separate passes) of this very command at HEAD, calibrated on a known byte count
    (tools/pmc_calib.hip) as /opt/skills/guides/MI355X_MICROARCH.md prescribes.  None when the summary is absent."""
    if not os.path.exists(PMC_SUMMARY):
        return None
    d = json.load(open(PMC_SUMMARY))
    k = d.get("kernels", {}).get(key)
    if not k:
        return None
    per_walker = k["bytes_per_launch"] / d["walkers"]
    return {"bytes_per_launch": per_walker * walkers, "bytes_per_walker": per_walker, "fetch_calibration": d["calibration"]["applied_fetch_factor"],
            "write_calibration": d["calibration"]["applied_write_factor"], "measured_at_walkers": d["walkers"], "source": "profiles/" + os.path.basename(PMC_SUMMARY)}


def kernel_stats():
    """Average launch durations (us) of the kernel classes named in the committed rocprofv3 --stats summary of this command."""
    out = {}
    if not os.path.exists(KERNEL_STATS):
        return out
    import csv

    for row in csv.DictReader(open(KERNEL_STATS)):
        name = row.get("kernel", "")
        for key, pat in (("k_orb1", "k_orb<1,"), ("k_orb5", "k_orb<5,"), ("k_step_lw", "k_step_lw<"), ("k_flush_lw", "k_flush_lw<"), ("k_sweep_res", "k_sweep_res<"), ("k_sweep_r8", "k_sweep_r8<"), ("k_kinetic_lw", "k_kinetic_lw<")):
            if pat in name and key not in out:
                out[key] = float(row["avg_us"])
    return out


def build_wf(device):
    import numpy as np

    import pyqmc_amd as pa

    mol = pa.systems.water_cluster()  # 24 atoms, 64 electrons, 184 AOs
    mf = pa.systems.random_mf(mol)
    wf = pa.generate_wf(mol, mf, device=device)
    rng = np.random.default_rng(11)
    a = 0.05 * rng.standard_normal((mol.natm, 4, 2))
    b = 0.05 * rng.standard_normal((4, 3))
    b[0] = [-0.25, -0.5, -0.25]
    wf.parameters["wf2acoeff"] = a
    wf.parameters["wf2bcoeff"] = b
    return mol, mf, wf


def socket_cores():
    """One logical CPU per physical core of socket 0 (`lscpu -p`), CPU model name."""
    import subprocess

    cpus, seen, model = [], set(), "unknown"
    try:
        for line in subprocess.run(["lscpu", "-p=CPU,CORE,SOCKET"], capture_output=True, text=True, check=True).stdout.splitlines():
            if line.startswith("#"):
                continue
            cpu, core, sock = (int(x) for x in line.split(",")[:3])
            if sock == 0 and core not in seen:
                seen.add(core)
                cpus.append(cpu)
        for line in subprocess.run(["lscpu"], capture_output=True, text=True, check=True).stdout.splitlines():
            if line.startswith("Model name"):
                model = line.split(":", 1)[1].strip()
    except Exception:
        cpus = list(range(os.cpu_count() or 1))
    usable = set(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else set(cpus)
    return [c for c in cpus if c in usable] or sorted(usable), model


def cgroup_cpu_quota():
    """CPUs the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def cpu_baseline(walkers_per_core, tstep, nsteps=2, max_procs=0):
    """SURVEY 8(d) / BASELINE.md section 3: the NumPy oracle (reference algorithm and structure: two gradient_value calls per
    move, per-(electron, atom) ECP loop, energy after every sweep; the reference's own timers mc.py:114-152 define what is
    counted) in P concurrent single-thread processes, P = physical cores of one socket, each pinned to its core with its own
    `walkers_per_core` walkers — the reference's own scaling model (vmc_parallel, mc.py:156-173: independent workers).  All
    workers start together; throughput = total walker-steps / (last end - first start)."""
    import multiprocessing as mp

    from oracle import baseline_worker

    cpus, model = socket_cores()
    socket = len(cpus)
    quota = cgroup_cpu_quota()
    if quota and quota < len(cpus):  # a container limited to fewer CPUs than the socket has: more processes would only time-slice
        cpus = cpus[: int(quota)]
    if max_procs > 0:
        cpus = cpus[:max_procs]
    P = len(cpus)
    ctx = mp.get_context("spawn")  # never fork a process that has HIP / RCCL state

    def leg(backend):
        start_at = time.time() + 20.0 + 0.05 * P  # imports + wave-function set-up of every worker happen before this
        with ctx.Pool(P) as pool:
            res = pool.map(baseline_worker.run, [(i, walkers_per_core, nsteps, tstep, cpus[i], start_at, backend) for i in range(P)], chunksize=1)
        t_begin, t_end = min(r[1] for r in res), max(r[2] for r in res)
        per_core = [r[3] / (r[2] - r[1]) for r in res]
        return {"value": sum(r[3] for r in res) / (t_end - t_begin), "per_core": float(sum(per_core) / P),
                "ao_share_of_wall_time": float(sum(r[4] for r in res) / sum(r[2] - r[1] for r in res)),
                "walker_steps": sum(r[3] for r in res), "seconds": t_end - t_begin, "late": max(r[1] for r in res) - start_at}

    c_leg, np_leg = leg("c"), leg("numpy")
    return {"value": c_leg["value"], "unit": "walker-steps/s", "cores": P, "kind": "port", "ao_backend": "c (oracle/ao_eval.c, gcc -O3, single thread per process)",
            "per_core": c_leg["per_core"], "ao_share_of_wall_time": c_leg["ao_share_of_wall_time"],
            "cpu_model": model, "socket_physical_cores": socket, "cgroup_cpu_quota": quota,
            "socket_extrapolated": c_leg["per_core"] * socket,
            "compiled_ao": {k: c_leg[k] for k in ("value", "per_core", "ao_share_of_wall_time")},
            "numpy_ao": {k: np_leg[k] for k in ("value", "per_core", "ao_share_of_wall_time")},
            "sample": f"{P} concurrent single-thread processes (one per physical core of socket 0, pinned"
                      + (f"; the container's cgroup allows {quota:g} CPUs of the socket's {socket} cores, so P = {P}: `socket_extrapolated` = per-core rate x {socket} "
                         "assumes the reference's own perfect worker scaling" if quota and quota < socket else "")
                      + f"), each {walkers_per_core} walkers x "
                      f"{nsteps} steps of the same sweep + energy evaluation: {c_leg['walker_steps']} walker-steps in {c_leg['seconds']:.1f} s "
                      f"(latest worker started {c_leg['late']:.2f} s after the common start); NumPy oracle with the AO routine compiled "
                      f"(`compiled_ao`, the headline: the reference's default AO back end is compiled code) and, as a second leg, in NumPy (`numpy_ao`); OMP/MKL threads = 1"}


def extra_measurements(pa, wf, dev, mol, W, args):
    """SURVEY 8(d) reporting grid, outside the timed region of `value`: sweep-only (the reference's "move time") next to
    sweep + energy at the bench's walker count, and sweep + energy at W/GPU in {4096, 16384, 65536}."""
    import numpy as np

    seeds = iter(range(5000, 6000))  # a fresh random stream per call: replaying one spreads the walkers unphysically (and makes the energy cheaper)

    def rate(walkers, energy, steps=4):
        if walkers != dev.W:
            wf.recompute(pa.initial_guess(mol, walkers, rng=np.random.default_rng(77)))
            dev.vmc_sweeps(args.tstep, 4, seed=next(seeds), energy=energy)  # leave the initial guess behind
        dev.vmc_sweeps(args.tstep, 1, seed=next(seeds), energy=energy)
        dev.sync()
        if not args.no_profile:
            dev.profile_enable(True)
        t0 = time.perf_counter()
        dev.vmc_sweeps(args.tstep, steps, seed=next(seeds), energy=energy)
        dev.sync()
        dt = time.perf_counter() - t0
        res = {"walker_steps_per_s": walkers * steps / dt, "ms_per_step": 1e3 * dt / steps}
        if not args.no_profile:
            launches, ms, pc = dev.profile_query()
            dev.profile_enable(False)
            if launches == steps and pc == launches * walkers * dev.N * 5:
                # the resident sweep (k_sweep_res: one launch per sweep, every launch bracketed): the whole sweep's MFMA-eligible and
                # AO-phase flops over the kernel's own duration, against the fp64 pipe both share
                tf = pc * 2.0 * 184 * 32 / (ms * 1e-3) / 1e12
                tv = (pc / 5.0) * (30 * 328 + 4 * 5 * 184) / (ms * 1e-3) / 1e12
                res.update(sweep="resident (k_sweep_r8 / k_sweep_res, one launch per sweep)", sweep_kernel_ms=ms / launches,
                           sweep_kernel_mfma_tflops=tf, sweep_kernel_mfma_frac=tf / FP64_MFMA_PEAK_TFLOPS,
                           sweep_kernel_ao_valu_frac=tv / FP64_MFMA_PEAK_TFLOPS, us_per_move_of_a_16_walker_block=1e3 * ms / launches / dev.N / max(1.0, walkers / 4096.0))
            else:
                res.update(sweep="one k_orb + k_step launch per move")
        return res

    out = {"sweep_only": rate(W, False), "sweep_plus_energy": rate(W, True), "by_walkers_per_gpu": {}, "sweep_only_by_walkers_per_gpu": {}}
    for w in (1024, 2048, 4096, 8192, 16384, 32768, 65536):
        out["by_walkers_per_gpu"][str(w)] = out["sweep_plus_energy"] if w == W else rate(w, True)
        out["sweep_only_by_walkers_per_gpu"][str(w)] = out["sweep_only"] if w == W else rate(w, False)
    return out


def rank_table(torch, dist, rank, local_rank, world):
    """One line per rank for the JSON: which device each process of the job really sits on, and the communicator's size as the
    process group reports it (so an 8-GPU record shows 8 distinct devices over RCCL, not 8 processes on one)."""
    if torch.cuda.is_available():
        p = torch.cuda.get_device_properties(local_rank)
        bus = None
        if all(hasattr(p, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            bus = f"{p.pci_domain_id:04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}"
        me = {"rank": rank, "local_rank": local_rank, "device": p.name, "gcn_arch": getattr(p, "gcnArchName", None), "pci_bus_id": bus,
              "hbm_gb": round(p.total_memory / 2**30, 1), "compute_units": p.multi_processor_count}
    else:  # (control-flow tests of the multi-rank bookkeeping on a CPU box)
        me = {"rank": rank, "local_rank": local_rank, "device": "cpu", "gcn_arch": None, "pci_bus_id": None, "hbm_gb": 0.0, "compute_units": 0}
    if dist is None:
        return {"rccl_ranks": 1, "backend": None, "ranks": [me]}
    table = [None] * world
    dist.all_gather_object(table, me)
    return {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "ranks": table}


def local_walkers(args, rank, world, weak_default, strong_default):
    """Walkers of this rank and the `scaling` label: weak = --walkers (or the config's per-GPU count) on every GPU; strong =
    --walkers (or the config's BASELINE total) split over the GPUs like the reference's configs.split (coord.py:72-80)."""
    from pyqmc_amd.dist import shard_bounds

    if args.scaling == "strong":
        total = args.walkers if args.walkers > 0 else strong_default
        lo, hi = shard_bounds(total, world)[rank]
        return hi - lo, total
    W = args.walkers if args.walkers > 0 else weak_default
    return W, W * world


def c4_bench(args, torch, dist, rank, local_rank, world, red_dev, fence):
    """--mode c4: BASELINE config 4 (H2O, 50 determinants x 2-body x 3-body Jastrow, VMC; 16384 walkers over 8 GPUs) — the
    wave-per-walker kernels.  Same timing contract as the headline; per block one all-reduce of the energy sums."""
    import numpy as np

    import pyqmc_amd as pa
    from pyqmc_amd.dist import allreduce_block

    W, total = local_walkers(args, rank, world, 2048, 16384)
    mol = pa.systems.water()
    mf = pa.systems.random_mf(mol, nvirt=8)
    wf = pa.generate_wf(mol, mf, determinants=pa.systems.random_determinants(mol, mf, 50), jastrow3=True, device=local_rank)
    wf.parameters["wf3ccoeff"] = 0.05 * np.random.default_rng(2).standard_normal(wf.parameters["wf3ccoeff"].shape)
    dev = wf.fused_device()
    wf.recompute(pa.initial_guess(mol, W, rng=np.random.default_rng(1234 + rank)))
    seed = 20260928 + 7919 * rank
    _, en_w, _ = dev.vmc_sweeps(args.tstep, max(args.warmup, 1), seed=seed, energy=True)
    allreduce_block(en_w.sum(axis=0) * W, en_w.shape[0] * W, device=red_dev)  # (warm-up of the reduction path too: its first call initialises the device-side ops)
    fence()
    t0 = time.perf_counter()
    acc, en, _ = dev.vmc_sweeps(args.tstep, args.steps, seed=seed + 1, energy=True)
    e_mean = allreduce_block(en.sum(axis=0) * W, en.shape[0] * W, device=red_dev)[0]
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    info = rank_table(torch, dist, rank, local_rank, world)
    if rank == 0:
        print(json.dumps({
            "metric": "walker-steps/sec (VMC sweep, H2O 50-determinant Slater x 2-body x 3-body Jastrow)", "value": total * args.steps / elapsed,
            "unit": "walker-steps/s", "n_gpus": world, "steps": args.steps, "warmup": max(args.warmup, 1), "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config C4: H2O ccECP-shaped tables, 8 e-, 50 determinants, 2-body (na=4, nb=4) and 3-body Jastrow; "
                                   "one step = 8 single-electron moves + EnergyAccumulator per walker",
                       "walkers_per_gpu": W, "global_walkers": total, "tstep": args.tstep, "parallelism": f"walker-sharded x{world}"},
            "acceptance": float(np.mean(acc)), "energy_total_mean": float(np.real(e_mean[5])), **info}), flush=True)


def dmc_bench(args, torch, dist, rank, local_rank, world, red_dev, fence):
    """--mode dmc: BASELINE config C5 (diamond 2x2x2 supercell, 64 e-, 8 k-points, tstep 0.02, T-moves, Ewald) — blocks of 5
    fused DMC steps (pqa_dmc_steps) followed by the block reduction and the distributed stochastic comb, whose walker
    exchange (only re-assigned walkers, device buffers over RCCL) is INSIDE the timed region."""
    import numpy as np

    import pyqmc_amd as pa
    from pyqmc_amd import dist as pdist
    from pyqmc_amd import pbc
    from pyqmc_amd.dmc import dmc_propagate

    W, total = local_walkers(args, rank, world, 4096, 32768)  # C5: 32768 walkers over 8 GPUs
    sup = pbc.get_supercell(pa.systems.diamond_primitive(), 2.0 * np.eye(3))
    wf = pa.generate_wf(sup, pbc.random_kmf(sup), device=local_rank)
    dev = wf.fused_device()
    np.random.seed(1234 + rank)
    cfg = pa.initial_guess(sup, W, rng=np.random.default_rng(99 + rank))
    acc = {"energy": pa.EnergyAccumulator(sup)}
    wf.recompute(cfg)
    dev.vmc_sweeps(0.3, 2, seed=7 + rank, energy=False)
    cfg.configs[...] = dev.configs()
    cfg.wrap += dev.wrap_delta()
    wf.recompute(cfg)
    en = np.real(acc["energy"](cfg, wf)["total"])
    (m1, m2), _ = pdist.allreduce_block([en.sum(), (en**2).sum()], len(en), device=red_dev)
    eref, esig = float(m1), float(np.sqrt(max(m2 - m1 * m1, 0.0)))
    weights = np.ones(W)
    nsb = 5
    stats = {"moved": 0, "bytes": 0, "blocks": 0, "gather_s": 0.0, "exchange_s": 0.0, "state_s": 0.0}
    wrng = np.random.default_rng(4242 + rank)

    def block(current):
        nonlocal cfg, weights
        blk, cfg, weights = dmc_propagate(wf, cfg, weights, 0.02, 10 * esig, eref, eref, nsteps=nsb, accumulators=acc, state_current=current)
        pdist.allreduce_block([blk["energytotal"] * blk["weight"] * W, blk["weight"] * W], W, device=red_dev)
        if args.unbalance > 0:  # rehearsal: the shards' total weights pushed apart so that the comb re-assigns walkers across ranks
            weights = weights * (1.0 + args.unbalance * (rank % 2)) * np.exp(0.3 * wrng.standard_normal(len(weights)))
        cfg, weights, info, _ = pdist.branch_distributed(cfg, weights, dev=dev, device_buffers=True if args.device_buffers else None)
        stats["moved"] += info["walkers moved"]; stats["bytes"] += info["bytes exchanged"]; stats["blocks"] += 1
        stats["gather_s"] += info["gather seconds"]; stats["exchange_s"] += info["exchange seconds"]; stats["state_s"] += info["state seconds"]
        return blk

    for i in range(max(args.warmup, 1)):
        block(i > 0)
    stats.update(moved=0, bytes=0, blocks=0, gather_s=0.0, exchange_s=0.0, state_s=0.0)
    nblocks = max((args.steps + nsb - 1) // nsb, 1)
    fence()
    t0 = time.perf_counter()
    for _ in range(nblocks):
        blk = block(True)
    fence()
    elapsed = time.perf_counter() - t0
    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())
    tbytes = torch.tensor([float(stats["bytes"])], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tbytes)
    state_err = None
    if args.check_state:  # after the exchanges: the resident state (kept walkers gathered, arrivals recomputed) against a fresh recompute
        logv = dev.value()[1]
        chk = torch.tensor([float(np.max(np.abs(dev.recompute(dev.configs())[1] - logv)))], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(chk, op=dist.ReduceOp.MAX)
        state_err = float(chk.item())
    info = rank_table(torch, dist, rank, local_rank, world)
    if rank == 0:
        steps = nblocks * nsb
        print(json.dumps({
            "metric": "walker-steps/sec (DMC, diamond 2x2x2 supercell 64e- Slater-Jastrow, tstep 0.02)", "value": total * steps / elapsed,
            "unit": "walker-steps/s", "n_gpus": world, "steps": steps, "warmup": max(args.warmup, 1) * nsb, "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE config C5: diamond 2x2x2 supercell (16 atoms, 64 e-, 8 k-points, ccECP-shaped synthetic tables), DMC "
                                   "tstep 0.02 with T-moves and Ewald energies; blocks of 5 steps + block reduction + distributed stochastic comb",
                       "walkers_per_gpu": W, "global_walkers": total, "parallelism": f"walker-sharded x{world}"},
            **info, "energy_total": float(blk["energytotal"]), "acceptance": float(blk["acceptance"]), "tmove_acceptance": float(blk["tmove_acceptance"]),
            "state_vs_recompute": state_err,
            "branching": {"blocks": stats["blocks"], "walkers_moved_per_block": stats["moved"] / max(stats["blocks"], 1),
                          "bytes_sent_per_block_rank0": stats["bytes"] / max(stats["blocks"], 1),
                          "bytes_sent_per_block_all_ranks": float(tbytes.item()) / max(stats["blocks"], 1),
                          "ms_per_block_rank0": {"weights_all_gather_and_comb": 1e3 * stats["gather_s"] / max(stats["blocks"], 1),
                                                 "pack_and_point_to_point": 1e3 * stats["exchange_s"] / max(stats["blocks"], 1),
                                                 "state_gather_and_recompute_of_arrivals": 1e3 * stats["state_s"] / max(stats["blocks"], 1)},
                          "unbalance": args.unbalance, "device_buffers": bool(args.device_buffers) or args.backend == "nccl",
                          "exchange": "all-gather of weights + point-to-point coordinates of re-assigned walkers only (RCCL)"}}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--walkers", type=int, default=0, help="walkers per GPU (weak scaling; default 65536 for the headline, the measured throughput "
                    "optimum; 4096 / 2048 for --mode dmc / c4) or in total (--scaling strong; default 32768 / 16384 for dmc / c4)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="weak: fixed walkers per GPU (the headline contract); strong: "
                    "a fixed ensemble split over the GPUs (BASELINE configs C4: 16384 and C5: 32768 walkers in total)")
    ap.add_argument("--tstep", type=float, default=0.3)
    ap.add_argument("--settle", type=int, default=30, help="untimed settling steps before the warm-up steps (clock ramp of a fresh process)")
    ap.add_argument("--mode", default="vmc", choices=["vmc", "dmc", "c4"], help="vmc: the headline metric (default); dmc: config C5 with branching; c4: config C4 (multi-determinant VMC)")
    ap.add_argument("--cpu-walkers", type=int, default=256, help="walkers per CPU-baseline process")
    ap.add_argument("--cpu-procs", type=int, default=0, help="CPU-baseline processes (0 = physical cores of one socket)")
    ap.add_argument("--no-extra", action="store_true", help="skip the sweep-only and walker-count grid measurements (extra)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the orbital kernel with HIP events")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for control-flow tests)")
    ap.add_argument("--same-gpu", action="store_true", help="TEST ONLY: all ranks use GPU 0 (needs --backend gloo)")
    ap.add_argument("--unbalance", type=float, default=0.0, help="--mode dmc rehearsal: odd ranks' weights scaled by 1 + this before every comb (and all "
                    "weights spread log-normally), so that walkers cross ranks in every block")
    ap.add_argument("--check-state", action="store_true", help="--mode dmc: after the timed blocks compare every rank's resident state with a fresh "
                    "recompute (max |log Psi| difference over all ranks -> state_vs_recompute)")
    ap.add_argument("--device-buffers", action="store_true", help="--mode dmc under gloo: pack / unpack the exchanged walkers in GPU tensors and hand the "
                    "library device pointers (the RCCL code path with gloo as the transport only)")
    args = ap.parse_args()

    for v in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OPENBLAS_NUM_THREADS"):
        os.environ.setdefault(v, "1")
    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.same_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.backend, rank=rank, world_size=world)

    import __graft_entry__ as ge

    if rank == 0:
        ge.build()
    if dist is not None:
        dist.barrier()
    import pyqmc_amd as pa

    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.mode in ("dmc", "c4"):
        (dmc_bench if args.mode == "dmc" else c4_bench)(args, torch, dist, rank, local_rank, world, red_dev, fence)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    mol, mf, wf = build_wf(local_rank)
    dev = wf.fused_device()
    W, total_walkers = local_walkers(args, rank, world, 65536, 65536 * world)
    cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1234 + rank))
    wf.recompute(cfg)
    seed = 20260928 + 7919 * rank

    red_dev = f"cuda:{local_rank}" if args.backend == "nccl" else "cpu"

    def fence():
        dev.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_block(en):
        """per-block energy accumulation across ranks (the path's only exchange step; RCCL all-reduce of 7 fp64)"""
        from pyqmc_amd.dist import allreduce_block

        return allreduce_block(en.sum(axis=0) * W, en.shape[0] * W, device=red_dev)[0]

    # untimed settling steps before the W warm-up steps: the first launches of a process load code objects and find the
    # clocks low (an evidence run of this round timed 37.3 ms per step right after 2 warm-up steps and 34.6 a minute later),
    # and the walkers leave the initial guess (33.2 ms per step after 3 settling steps, 33.0 after 30, 90 or 200)
    dev.vmc_sweeps(args.tstep, args.settle, seed=seed + 7, energy=True)
    if not args.no_profile:
        dev.profile_enable(True)  # during the warm-up too: the event pairs are created there, not inside the timed region
    if args.warmup > 0:
        _, en_w, _ = dev.vmc_sweeps(args.tstep, args.warmup, seed=seed, energy=True)
        reduce_block(en_w)  # also warms torch's allocator / RCCL communicator outside the timed region
    if not args.no_profile:
        dev.profile_enable(True)  # reset the accounting; the events stay
    fence()
    t0 = time.perf_counter()
    dev.timer_start()
    acc, en, _ = dev.vmc_sweeps(args.tstep, args.steps, seed=seed + 1, energy=True)
    ev_ms = dev.timer_stop()
    e_mean = reduce_block(en)
    fence()
    elapsed = time.perf_counter() - t0
    launches, orb_ms, point_comps = (0, 0.0, 0.0) if args.no_profile else dev.profile_query()
    c_launches, c_ms = (0, 0.0) if args.no_profile else dev.profile_query_commit()
    p_launches, p_ms, p_groups = (0, 0.0, 1) if args.no_profile else dev.profile_query_part()
    if not args.no_profile:
        dev.profile_enable(False)
    ecp_pts = dev.last_ecp_points()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    info = rank_table(torch, dist, rank, local_rank, world)
    if rank == 0:
        nao, nmo = 184, 32
        value = total_walkers * args.steps / elapsed
        out = {
            "metric": "walker-steps/sec (VMC sweep, 64e- Slater-Jastrow)",
            "value": value, "unit": "walker-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "(H2O)8 cluster: 64 e- (32 up, 32 dn), 24 atoms, 184 AOs (ccECP cc-pVDZ-shaped synthetic "
                                   "tables), 1 determinant, 2-body Jastrow (na=4, nb=4), ccECP-shaped ECP threshold=10; "
                                   "one step = 64 single-electron moves + EnergyAccumulator per walker",
                       "walkers_per_gpu": W, "global_walkers": total_walkers, "tstep": args.tstep, "parallelism": f"walker-sharded x{world}"},
            "acceptance": float(np.mean(acc)), "energy_total_mean": float(e_mean[5]),
            "ecp_points_per_walker_step": ecp_pts / W, "stream_event_ms": ev_ms, **info,
        }
        resident = (not args.no_profile) and launches == args.steps and point_comps == launches * W * 64 * 5  # one bracketed launch per sweep
        if resident:
            # The dominant kernel since round 6 is the resident sweep k_sweep_r8 (one launch per sweep: AO evaluation, contraction, both
            # Jastrow evaluations, Metropolis test and Sherman-Morrison of all 64 moves on chip).  Its bound is the fp64 pipe, which the
            # matrix and the vector path share on this part: `achieved` = the USEFUL fp64 flops of a sweep (SURVEY 8(d) formula sheet:
            # per move F_ao(5) + 2*5*M*n contraction + 16 n ratio sums + 2 F_j2 + acc * 4 n^2 Sherman-Morrison) over the kernel's own
            # duration (HIP events around every launch on the library's stream); the matrix part alone is `mfma_frac`.
            n_s_, acc_ = 32, float(np.mean(acc))
            f_ao5_ = 30 * 328 + 4 * 5 * nao
            f_move_ = f_ao5_ + 2 * 5 * nao * n_s_ + 2 * (2 * 4 * n_s_) + 2 * 9500.0 + acc_ * 4 * n_s_ * n_s_
            moves = point_comps / 5.0
            achieved = moves * f_move_ / (orb_ms * 1e-3) / 1e12
            mf_ = point_comps * 2.0 * nao * nmo / (orb_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_sweep_r8 (resident sweep: GTO AO evaluation + AO->MO fp64 MFMA contraction + Jastrow pair sums + Metropolis "
                                                          "+ Sherman-Morrison of all 64 moves of 8 walkers per block, two blocks per CU, one launch per sweep)",
                               "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                               "peak_note": "fp64 matrix peak = fp64 vector peak on MI355X, one shared pipe (tools/ubench.hip, tools/scratch/mfma_probe.hip)",
                               "mfma_tflops": mf_, "mfma_frac": mf_ / FP64_MFMA_PEAK_TFLOPS,
                               "ao_valu_frac": moves * f_ao5_ / (orb_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                               "jastrow_valu_frac": moves * 2 * 9500.0 / (orb_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                               "useful_flop_per_move": f_move_, "traffic": pmc_kernel("k_sweep_r8", W),
                               "algorithmic_bytes_per_launch": W * (2 * 8 * (3 * 64 + 2 * n_s_ * n_s_) + 64 * (8 * (4 * n_s_ + 4) + acc_ * (5 * nmo * 8 + 1))),
                               "launches": launches, "avg_launch_ms": orb_ms / launches, "launches_timed": "every launch (one per sweep)",
                               "us_per_move_of_a_16_walker_pair_of_blocks": 1e3 * (orb_ms / launches) / 64 / (W / 4096.0),
                               "kernel_share_of_step": (orb_ms / launches) * args.steps / (1e3 * elapsed)}
        elif not args.no_profile and launches:
            flops = point_comps * 2.0 * nao * nmo  # AO->MO contraction only: the MFMA-eligible work of the kernel
            achieved = flops / (orb_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_orb (fused GTO AO evaluation + AO->MO fp64 MFMA contraction)",
                               "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                               "peak_achievable": FP64_MFMA_ACHIEVABLE_TFLOPS, "frac_of_achievable": achieved / FP64_MFMA_ACHIEVABLE_TFLOPS,
                               "traffic": pmc_kernel("k_orb5", W),
                               "algorithmic_bytes_per_launch": (5 * nmo * 8 + 24) * (point_comps / launches / 5),
                               "launches": launches, "avg_launch_ms": orb_ms / launches,
                               "launches_timed": "1 in 4 of the step's 64 move launches (an event pair costs ~2 us of stream time)",
                               "kernel_share_of_step": (orb_ms / launches) * 64 * args.steps / (1e3 * elapsed),
                               "flops_per_point_component": 2 * nao * nmo}
        n_s, N = 32, 64
        kb = int(os.environ.get("PQA_LW_KB", "-1"))
        kb = (5 if n_s >= 24 else 4) if kb < 0 else (n_s if kb == 0 else min(kb, n_s))  # the library's default (lw_setup)
        if not args.no_profile and p_launches:
            # The Jastrow distance sweep of north_star lives in k_step_lw (one launch per move: decide electron e, propose e + 1).
            # ALGORITHMIC bytes per walker and launch: the coordinates of all electrons once (N x 24 B), the proposal's value and
            # gradient rows (4 n x 8 B) and inverse row e (n x 8), the KB - 1 other inverse rows of the electron block read and
            # written + row e written, the update vectors V, R to the block buffers (2 n x 8), the cached rows (4 n x 8) and
            # inverse row (n x 8) of electron e + 1, and ~100 B of per-walker scalars (proposal, drift, selector, flags).
            # `achieved` prices those against the event-measured launch time; `traffic` is what the HBM counters saw.
            alg = N * 24 + 4 * n_s * 8 + n_s * 8 + 2 * (kb - 1) * n_s * 8 + n_s * 8 + 2 * n_s * 8 + 4 * n_s * 8 + n_s * 8 + 100
            ach = alg * W / (p_ms / p_launches * 1e-3) / 1e9
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_step_lw (one launch per move: Slater ratio sums + Jastrow e-e / e-ion distance sums at the proposal, "
                                                              "Metropolis test, Sherman-Morrison rows of the electron block, then the same sums and the proposal of the next electron)",
                                   "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": ach / HBM_ACHIEVABLE_GBS,
                                   "traffic": pmc_kernel("k_step_lw", W), "launches": p_launches, "avg_launch_ms": p_ms / p_launches,
                                   "kernel_share_of_step": (p_ms / p_launches) * N * args.steps / (1e3 * elapsed),
                                   "algorithmic_bytes_per_walker": alg, "thread_groups_per_walker": p_groups}
        if not args.no_profile and c_launches:
            # k_flush_lw, the deferred half of the blocked Sherman-Morrison update: after every block of KB moves of a spin it
            # carries the n - KB rows outside the block through HBM (read + write) plus the block's KB update-vector pairs.
            # Walkers are interleaved in every cache line, so ALL walkers' rows cross HBM: algorithmic bytes per walker.
            alg = 2 * 8 * (n_s - kb) * n_s + 2 * 8 * kb * n_s  # = 16 n_s^2 whatever the block size
            ach = alg * W / (c_ms / c_launches * 1e-3) / 1e9
            flushes_per_step = 2 * -(-n_s // kb) if kb < n_s else 0
            out["roofline_hbm_flush"] = {"bound": "hbm", "kernel": "k_flush_lw (blocked Sherman-Morrison: rows outside the electron block, once per block of KB moves)",
                                         "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                         "peak_achievable": HBM_ACHIEVABLE_GBS, "frac_of_achievable": ach / HBM_ACHIEVABLE_GBS,
                                         "traffic": pmc_kernel("k_flush_lw", W), "launches": c_launches, "avg_launch_ms": c_ms / c_launches,
                                         "kernel_share_of_step": (c_ms / c_launches) * flushes_per_step * args.steps / (1e3 * elapsed),
                                         "algorithmic_bytes_per_walker": alg, "block_KB": kb}
        # ---- the AO phase of the same k_orb launches against the fp64 VECTOR peak (one pipe with the matrix
        # path on this part: the two fractions add up to the pipe's utilisation by useful flops).  F_ao = 30 P + 4 ncomp M per point
        # with P = 328 primitives (one exp each), M = 184 functions, ncomp = 5.
        nprim = 328
        f_ao5, f_ao1 = 30 * nprim + 4 * 5 * nao, 30 * nprim + 4 * 1 * nao
        if not args.no_profile and launches and "roofline" in out and not resident:
            pts = point_comps / 5.0
            ach = pts * f_ao5 / (orb_ms * 1e-3) / 1e12
            out["roofline_valu"] = {"bound": "valu", "kernel": "k_orb, AO phase (GTO value / gradient / Laplacian of 184 functions, 328 exp per point)",
                                    "achieved": ach, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach / FP64_MFMA_PEAK_TFLOPS,
                                    "flops_per_point": f_ao5, "exp_per_point": nprim,
                                    "pipe_frac_mfma_plus_valu": (ach + out["roofline"]["achieved"]) / FP64_MFMA_PEAK_TFLOPS,
                                    "note": "same launches and event times as `roofline`; SURVEY 8(d) formula F_ao = 30 P + 4 ncomp M"}
        stats_k = kernel_stats()
        if stats_k.get("k_orb1") and ecp_pts:
            pts1 = ecp_pts / 2.0  # one value-only launch per spin and energy evaluation
            us = stats_k["k_orb1"]
            out["roofline_orb1"] = {"bound": "mfma", "kernel": "k_orb<1> (orbital values at the ECP quadrature points, one launch per spin)",
                                    "achieved": pts1 * 2.0 * nao * nmo / (us * 1e-6) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": pts1 * 2.0 * nao * nmo / (us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                    "valu_frac": pts1 * f_ao1 / (us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                    "avg_launch_us": us, "points_per_launch": pts1,
                                    "source": "launch duration from profiles/" + os.path.basename(KERNEL_STATS) + " (rocprofv3 of this command), points of THIS run"}
        # ---- the whole step: useful flops per walker-step from the formula sheet (minimal count: one AO / MO evaluation per move, the
        # cached rows serve the old position and the kinetic energy) and counter bytes per walker-step, both over ms_per_step
        acc_rate = float(np.mean(acc))
        f_j2 = 9500.0  # Jastrow value + gradient of one electron (F_j2), twice per move; with the Laplacian once per electron in the energy
        f_move = f_ao5 + 2 * 5 * nao * n_s + 2 * (2 * 4 * n_s) + 2 * f_j2 + acc_rate * 4 * n_s * n_s
        f_kin = N * (2 * 5 * n_s + 1.3 * f_j2)
        f_ecp = (ecp_pts / W) * (f_ao1 + 2 * nao * n_s + 2 * n_s + 0.5 * f_j2)
        f_step = N * f_move + f_kin + f_ecp
        per_step = {"k_step_lw": 77, "k_flush_lw": 14, "k_kinetic_lw": 1, "k_orb5": 64, "k_orb1": 2, "k_ecp_point": 2, "k_ecp_count": 1, "k_ecp_fill": 1}
        if resident:
            per_step = {"k_sweep_r8": 1, "k_tile_draws": 1, "k_kinetic_lw": 1, "k_orb1": 2, "k_ecp_point": 2, "k_ecp_count": 1, "k_ecp_fill": 1}
        b_step = None
        if os.path.exists(PMC_SUMMARY):
            dpm = json.load(open(PMC_SUMMARY))
            if all(k in dpm.get("kernels", {}) for k in per_step):
                b_step = sum(n * dpm["kernels"][k]["bytes_per_launch"] / dpm["walkers"] for k, n in per_step.items())
        ms = 1e3 * elapsed / args.steps
        out["roofline_step"] = {"useful_flop_per_walker_step": f_step, "mfma_eligible_flop_per_walker_step": N * 2 * 5 * nao * n_s + (ecp_pts / W) * 2 * nao * n_s,
                                "achieved_tflops": f_step * W / (ms * 1e-3) / 1e12, "frac_of_fp64_peak": f_step * W / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                "counter_bytes_per_walker_step": b_step,
                                "achieved_gbs": None if b_step is None else b_step * W / (ms * 1e-3) / 1e9,
                                "frac_of_hbm_peak": None if b_step is None else b_step * W / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "survey_B_sweep_bytes": 2 * 8 * (3 * N + 2 * n_s * n_s) + N * 2 * 8 * (24 * 4 + 2 * 4),
                                "formula": "N (F_ao(5) + 2*5*M*n + 16 n + 2 F_j2 + acc*4 n^2) + N (10 n + 1.3 F_j2) + ecp_points (F_ao(1) + 2 M n + 2 n + F_j2/2); "
                                           "bytes: launches per step x calibrated counter bytes per launch (resident sweep: k_sweep_r8 1, k_tile_draws 1, k_orb<1> 2, kinetic, ECP passes; "
                                           "launch-per-move sweep: k_step_lw 77, k_flush_lw 14, k_orb 64 + 2, kinetic, ECP passes)"}
        if world == 1 and not args.no_extra:
            out["extra"] = extra_measurements(pa, wf, dev, mol, W, args)
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (the other ranks would just wait)
            out["cpu_baseline"] = cpu_baseline(args.cpu_walkers, args.tstep, max_procs=args.cpu_procs)
            out["speedup_vs_cpu_measured"] = value / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_socket_extrapolated"] = value / out["cpu_baseline"]["socket_extrapolated"]
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
