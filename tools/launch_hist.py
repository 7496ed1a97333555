"""Per-kernel launch histogram of the LAST frame in a rocprofv3 --kernel-trace database (bench.py --inflight 1): launches, summed
and longest duration, grids, and the time the device was busy inside the frame's window.

    rocprofv3 --kernel-trace -d /tmp/p -o b -- python bench.py --no-cpu-baseline --no-next-rows --inflight 1 --steps 3 --warmup 1
    python tools/launch_hist.py /tmp/p/*/b_results.db
"""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, start, end, grid_x, grid_y, workgroup_x from kernels order by start"))
# last frame only: find last k_ppf_matrix start
starts = [s for n, s, e, *_ in rows if 'k_ppf_matrix' in n]
t0 = starts[-1]
last = [(n, s, e, gx, gy, wx) for n, s, e, gx, gy, wx in rows if s >= t0 - 8e6]
print("kernels in window:", len(last), "window ms:", (last[-1][2] - last[0][1]) / 1e6)
busy = 0; cur_end = 0
for n, s, e, *_ in last:
    if s > cur_end: busy += e - s; cur_end = e
    elif e > cur_end: busy += e - cur_end; cur_end = e
print("busy ms", busy / 1e6)
agg = collections.OrderedDict()
for n, s, e, gx, gy, wx in last:
    k = n.split('(')[0][-60:]
    a = agg.setdefault(k, [])
    a.append(((e - s) / 1e3, gx, gy, wx))
for k, a in sorted(agg.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
    ds = [x[0] for x in a]
    print(f"{k:<62}{len(a):>5}{sum(ds):>10.1f} us  max {max(ds):>8.1f}  grids {sorted(set((x[1],x[2]) for x in a))[:4]}")
