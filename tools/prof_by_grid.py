"""Per-(kernel, grid size) duration summary of a rocprofv3 rocpd database: python tools/prof_by_grid.py db [substr]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
sub = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, d.grid_size_x, d.grid_size_y, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d join {ks} s "
     f"on d.kernel_id = s.id group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 6 desc")
for name, gx, gy, n, avg, tot in c.execute(q):
    if sub in name:
        print(f"{name[:60]:60s} grid=({gx},{gy}) calls={n} avg_us={avg:.1f} total_us={tot:.0f}")
