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


def pmc_traffic(point_comps_per_launch):
    """HBM bytes per k_orb launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
    in separate runs, profiles/r01_pmc_summary.json): measured bytes per (point, component) x this run's
    average launch size.  None if the summary is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
    if not os.path.exists(path):
        return None
    d = json.load(open(path))
    return {"bytes_per_launch": d["k_orb5_bytes_per_point_component"] * point_comps_per_launch,
            "algorithmic_bytes_per_launch": (32 * 8 + 24 / 5) * point_comps_per_launch, "source": "profiles/r01_pmc_summary.json"}


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


def cpu_baseline(walkers, tstep, nsteps=3):
    """The oracle (NumPy restatement of the reference algorithm, reference structure: two
    gradient_value calls per move, per-(electron, atom) ECP loop, energy after every sweep) timed on
    one host core for one step of `walkers` walkers; the reference's own timers (move + accumulator,
    mc.py:114-152) define what is counted."""
    import numpy as np

    import pyqmc_amd as pa
    from oracle import vmc as ovmc
    from tests import helpers

    mol = pa.systems.water_cluster()
    mf = pa.systems.random_mf(mol)
    owf = helpers.oracle_wf(mol, mf)
    rng = np.random.default_rng(5)
    cfg = pa.initial_guess(mol, walkers, rng=rng)
    N, necp = 64, mol.natm
    gauss, unif = rng.standard_normal((nsteps, N, walkers, 3)), rng.random((nsteps, N, walkers))
    rot = np.broadcast_to(np.eye(3), (nsteps, N, necp, 3, 3)).copy()
    eunif = rng.random((nsteps, N, necp, walkers))
    owf.recompute(cfg)  # set-up (the reference's vmc_worker also starts from a recompute) stays outside the clock
    t0 = time.perf_counter()
    ovmc.vmc_worker(mol, owf, cfg, tstep, gauss, unif, rot, eunif)
    secs = time.perf_counter() - t0
    return {"value": walkers * nsteps / secs, "unit": "walker-steps/s", "cores": 1, "kind": "port",
            "sample": f"{walkers} walkers x {nsteps} steps of the same sweep + energy evaluation ({secs:.1f} s), "
                      "NumPy oracle on one host core, OMP/MKL threads pinned to 1"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--walkers", type=int, default=65536, help="walkers per GPU (weak scaling); 65536 is the measured throughput optimum")
    ap.add_argument("--tstep", type=float, default=0.3)
    ap.add_argument("--cpu-walkers", type=int, default=512)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket the orbital kernel with HIP events")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for control-flow tests)")
    ap.add_argument("--same-gpu", action="store_true", help="TEST ONLY: all ranks use GPU 0 (needs --backend gloo)")
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

    mol, mf, wf = build_wf(local_rank)
    dev = wf.fused_device()
    W = args.walkers
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

    if args.warmup > 0:
        _, en_w, _ = dev.vmc_sweeps(args.tstep, args.warmup, seed=seed, energy=True)
        reduce_block(en_w)  # also warms torch's allocator / RCCL communicator outside the timed region
    if not args.no_profile:
        dev.profile_enable(True)
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
    if not args.no_profile:
        dev.profile_enable(False)
    ecp_pts = dev.last_ecp_points()

    tmax = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed = float(tmax.item())

    if rank == 0:
        nao, nmo = 184, 32
        total_walkers = W * world
        value = total_walkers * args.steps / elapsed
        out = {
            "metric": "walker-steps/sec (VMC sweep, 64e- Slater-Jastrow)",
            "value": value, "unit": "walker-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "(H2O)8 cluster: 64 e- (32 up, 32 dn), 24 atoms, 184 AOs (ccECP cc-pVDZ-shaped synthetic "
                                   "tables), 1 determinant, 2-body Jastrow (na=4, nb=4), ccECP-shaped ECP threshold=10; "
                                   "one step = 64 single-electron moves + EnergyAccumulator per walker",
                       "walkers_per_gpu": W, "global_walkers": total_walkers, "tstep": args.tstep, "parallelism": f"walker-sharded x{world}"},
            "acceptance": float(np.mean(acc)), "energy_total_mean": float(e_mean[5]),
            "ecp_points_per_walker_step": ecp_pts / W, "stream_event_ms": ev_ms,
        }
        if not args.no_profile and launches:
            flops = point_comps * 2.0 * nao * nmo  # AO->MO contraction only: the MFMA-eligible work of the kernel
            achieved = flops / (orb_ms * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": "k_orb (fused GTO AO evaluation + AO->MO fp64 MFMA contraction)",
                               "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": pmc_traffic(point_comps / launches),
                               "launches": launches, "avg_launch_ms": orb_ms / launches,
                               "launches_timed": "1 in 4 of the step's 64 move launches (an event pair per launch cost 4 % of the step)",
                               "kernel_share_of_step": (orb_ms / launches) * 64 * args.steps / (1e3 * elapsed),
                               "flops_per_point_component": 2 * nao * nmo}
        if not args.no_profile and c_launches:
            # The streaming kernel of the step: k_flush_lw, the deferred half of the blocked Sherman-Morrison update.  After
            # every block of KB moves of a spin it carries the n - KB rows outside the block through HBM once (read + write)
            # for every walker with an accepted move in the block, plus that walker's KB update-vector pairs (V, R).
            # Algorithmic bytes per such walker (n = 32, KB from the library's rule / PQA_LW_KB):
            n_s = 32
            kb = int(os.environ.get("PQA_LW_KB", "-1"))
            kb = 4 if kb < 0 else (n_s if kb == 0 else min(kb, n_s))
            bytes_walker = 2 * 8 * (n_s - kb) * n_s + 2 * 8 * kb * n_s
            a_ = float(np.mean(acc))
            touched = W * (1.0 - (1.0 - a_) ** kb)  # walkers with >= 1 accepted move among the block's KB (independent moves)
            ach = touched * c_launches * bytes_walker / (c_ms * 1e-3) / 1e9
            traffic = None
            path = os.path.join(ROOT, "profiles", "r01_pmc_summary.json")
            if os.path.exists(path):
                d = json.load(open(path)).get("k_flush_lw")
                if d:
                    traffic = {"bytes_per_launch": d["bytes_per_walker"] * W, "source": "profiles/r01_pmc_summary.json"}
            flushes_per_step = 2 * (n_s // kb) if kb < n_s else 0
            out["roofline_hbm"] = {"bound": "hbm", "kernel": "k_flush_lw (blocked Sherman-Morrison: rows outside the electron block, once per block of KB moves)",
                                   "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                   "traffic": traffic, "launches": c_launches, "avg_launch_ms": c_ms / c_launches,
                                   "kernel_share_of_step": (c_ms / c_launches) * flushes_per_step * args.steps / (1e3 * elapsed),
                                   "algorithmic_bytes_per_touched_walker": bytes_walker, "block_KB": kb}
        if not args.no_cpu_baseline and world == 1:  # reported at N = 1 only (the other ranks would just wait)
            out["cpu_baseline"] = cpu_baseline(args.cpu_walkers, args.tstep)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
