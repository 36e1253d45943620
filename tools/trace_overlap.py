"""Concurrency in a rocprofv3 --kernel-trace: per kernel family the launches, mean duration, and how much of the busy time had
two or more kernels in flight.  usage: python tools/trace_overlap.py <kernel_trace.csv> [last_fraction=0.3]"""
import csv, re, sys
from collections import defaultdict

path = sys.argv[1]
frac = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"<.*", "", r["Kernel_Name"].replace("void ", "")).split("(")[0]))
rows.sort()
t_lo = rows[0][0] + (1.0 - frac) * (rows[-1][1] - rows[0][0])
rows = [r for r in rows if r[0] >= t_lo]
ev = []
for s, e, n in rows:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort()
active = defaultdict(int)
busy = multi = 0
pair = defaultdict(int)
last = ev[0][0]
for t, d, n in ev:
    dt = t - last
    k = sum(active.values())
    if k >= 1: busy += dt
    if k >= 2:
        multi += dt
        pair[tuple(sorted(x for x in active if active[x] > 0))] += dt
    active[n] += d
    last = t
span = rows[-1][1] - rows[0][0]
print(f"window {span / 1e6:.2f} ms, busy {busy / 1e6:.2f} ms, >=2 kernels in flight {multi / 1e6:.2f} ms ({100.0 * multi / max(busy, 1):.1f} % of busy)")
st = defaultdict(lambda: [0, 0])
for s, e, n in rows:
    st[n][0] += 1; st[n][1] += e - s
for n, (c, tot) in sorted(st.items(), key=lambda x: -x[1][1])[:12]:
    print(f"  {n:28s} launches {c:6d}  mean {tot / c / 1e3:9.1f} us  total {tot / 1e6:8.2f} ms")
for p, dt in sorted(pair.items(), key=lambda x: -x[1])[:8]:
    print(f"  concurrent {' + '.join(p):50s} {dt / 1e6:8.2f} ms")
