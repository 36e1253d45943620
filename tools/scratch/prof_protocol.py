import cProfile, pstats, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np, helpers, pyqmc_amd as pa
W = int(sys.argv[1])
mol = pa.systems.water_cluster(); wf = pa.generate_wf(mol, pa.systems.random_mf(mol))
acc = {"energy": pa.EnergyAccumulator(mol)}
cfg = pa.initial_guess(mol, W, rng=np.random.default_rng(1))
helpers.protocol_vmc_worker(wf, cfg, 0.3, 1, acc)
pr = cProfile.Profile(); pr.enable(); helpers.protocol_vmc_worker(wf, cfg, 0.3, 2, acc); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
