"""Summarise a rocprofv3 rocpd database (kernel trace) as CSV: python tools/prof_stats.py db out.csv"""
import csv, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,total_calls,total_duration,average,percentage from top_kernels").fetchall()
w = csv.writer(open(sys.argv[2], "w", newline="") if len(sys.argv) > 2 else sys.stdout)
w.writerow(["kernel", "calls", "total_us", "avg_us", "pct"])
for r in rows:
    w.writerow([r[0][:110], r[1], round(r[2], 1), round(r[3], 2), round(r[4], 2)])
