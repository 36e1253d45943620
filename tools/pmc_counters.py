"""Per-kernel mean of a PMC counter from a rocprofv3 rocpd database: python tools/pmc_counters.py db [out.csv]"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection "
                 "group by kernel_name, counter_name order by sum(value) desc").fetchall()
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "counter", "dispatches", "mean_value_KiB", "sum_value_KiB", "mean_duration_ns"])
for r in rows:
    w.writerow([r[0][:100], r[1], r[2], round(r[3], 2), round(r[4], 1), round(r[5])])
