"""Aggregate the rocprofv3 PMC passes of tools/refresh_evidence.sh into one JSON (per kernel class: dispatches, counter bytes
per launch — raw and calibrated —, mean duration) that bench.py reads for its `traffic` fields.

    python tools/pmc_summary.py FETCH.db WRITE.db CALIB_FETCH.db CALIB_WRITE.db WALKERS OUT.json

Counters are collected exactly as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in separate
`--pmc` passes without trace options (the TCC block cannot hold both).  rocprofv3 reports them in KiB.  Calibration: the
guide says FETCH_SIZE is uncalibrated outside 16-B/lane streaming reads (where it reads 1/2), so tools/pmc_calib.hip moves a
known 2 GiB with this code's own patterns (8 B/lane coalesced reads / writes / read-modify-write) under the same counters;
the factor known / counter of `k_read8` and `k_write8` is applied to every kernel (they all stream doubles, walker or
point index fastest).
"""
import json
import sqlite3
import sys

CLASSES = {  # substring of the demangled kernel name -> key
    "k_sweep_r8": "k_sweep_r8", "k_sweep_res": "k_sweep_res", "k_tile_draws": "k_tile_draws", "k_orb<5": "k_orb5", "k_orb<1": "k_orb1", "k_orb_wide": "k_orb_wide", "k_step_lw": "k_step_lw", "k_flush_lw": "k_flush_lw",
    "k_kinetic_lw": "k_kinetic_lw", "k_transpose": "k_transpose", "k_cache_to_rc": "k_cache_to_rc", "k_cache_from_rc": "k_cache_from_rc",
    "k_ecp_point": "k_ecp_point", "k_ecp_count": "k_ecp_count", "k_ecp_fill": "k_ecp_fill", "k_pbc_prepass": "k_pbc_prepass", "k_ewald": "k_ewald",
}


def per_kernel(db):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, count(*), avg(value), sum(value), avg(duration) from counters_collection group by kernel_name").fetchall()
    return {r[0]: {"dispatches": r[1], "mean_kib": r[2], "sum_kib": r[3], "mean_ns": r[4]} for r in rows}


def main():
    fdb, wdb, cf, cw, walkers, out = sys.argv[1:7]
    walkers = int(walkers)
    calib = {}
    known = float(1 << 31)
    for name, d in per_kernel(cf).items():
        for k in ("k_read8", "k_read16", "k_rmw8"):
            if k in name:
                calib["fetch_" + k] = known / (d["mean_kib"] * 1024.0)
    for name, d in per_kernel(cw).items():
        for k in ("k_write8", "k_rmw8"):
            if k in name:
                calib["write_" + k] = known / (d["mean_kib"] * 1024.0)
    ff, wf = calib.get("fetch_k_read8", 1.0), calib.get("write_k_write8", 1.0)
    fetch, write = per_kernel(fdb), per_kernel(wdb)
    res = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace options) of `python bench.py --steps 2 "
                     "--warmup 1 --no-cpu-baseline --no-profile --no-extra`; tools/refresh_evidence.sh, tools/pmc_summary.py",
           "walkers": walkers,
           "calibration": {**calib, "applied_fetch_factor": ff, "applied_write_factor": wf,
                           "note": "known bytes / counter bytes of tools/pmc_calib.hip (2 GiB streamed with 8 B/lane coalesced accesses)"},
           "kernels": {}}
    tot_f = tot_w = 0.0
    for name, d in fetch.items():
        w = write.get(name, {"mean_kib": 0.0, "sum_kib": 0.0})
        tot_f += d["sum_kib"] * 1024 * ff
        tot_w += w["sum_kib"] * 1024 * wf
        for sub, key in CLASSES.items():
            if sub in name:
                e = res["kernels"].setdefault(key, {"dispatches": 0, "fetch_bytes_sum_raw": 0.0, "write_bytes_sum_raw": 0.0, "ns_sum": 0.0})
                e["dispatches"] += d["dispatches"]
                e["fetch_bytes_sum_raw"] += d["sum_kib"] * 1024
                e["write_bytes_sum_raw"] += w["sum_kib"] * 1024
                e["ns_sum"] += d["mean_ns"] * d["dispatches"]
    for key, e in res["kernels"].items():
        n = e["dispatches"]
        e["fetch_bytes_per_launch_raw"] = e.pop("fetch_bytes_sum_raw") / n
        e["write_bytes_per_launch_raw"] = e.pop("write_bytes_sum_raw") / n
        e["bytes_per_launch"] = e["fetch_bytes_per_launch_raw"] * ff + e["write_bytes_per_launch_raw"] * wf
        e["mean_launch_us_under_pmc"] = e.pop("ns_sum") / n * 1e-3
    res["total_bytes_all_kernels"] = tot_f + tot_w
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: round(v["bytes_per_launch"] / walkers, 1) for k, v in res["kernels"].items()}), "B per walker and launch;", calib)


if __name__ == "__main__":
    main()
