#!/usr/bin/env python
"""Turns a rocprofv3 results database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`, rocpd/sqlite output of
ROCm 7.2) into the small text summary that is committed under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r01/bench_results.db profiles/r01_kernel_stats.txt "command line"
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::trampoline_kernel<.*?radix_sort_onesweep_(\w+)<.*", r"rocprim::radix_sort_onesweep_\1<...>", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    db, out = sys.argv[1], sys.argv[2]
    cmd = sys.argv[3] if len(sys.argv) > 3 else ""
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    agg = {}
    for name, calls, total, avg, pct in rows:
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 0.0])
        a[0] += calls
        a[1] += total
        a[2] += pct
    with open(out, "w") as f:
        f.write(f"# rocprofv3 --kernel-trace --stats summary (total in ms, average in us)\n# command: {cmd}\n")
        f.write(f"{'kernel':<112}{'calls':>8}{'total_ms':>14}{'avg_us':>12}{'pct':>8}\n")
        for k, (calls, total, pct) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            f.write(f"{k:<112}{calls:>8}{total / 1e3:>14.1f}{total / calls:>12.1f}{pct:>8.2f}\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
