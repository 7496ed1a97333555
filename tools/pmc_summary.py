#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in separate runs, as
/opt/skills/guides/MI355X_MICROARCH.md prescribes).  FETCH_SIZE / WRITE_SIZE are in KiB (x1024); on gfx950 FETCH_SIZE
under-reports wide coalesced reads by 2x (same guide), so the read side is reported both raw and doubled.

    python tools/pmc_summary.py gpurun_out/pmc_fetch/b_results.db gpurun_out/pmc_write/b_results.db profiles/pmc_latest.json
"""
import json
import sqlite3
import sys


def per_kernel(db, counter):
    cur = sqlite3.connect(db).cursor()
    out = {}
    for name, value, dur in cur.execute(
            "select kernel_name, value, duration from counters_collection where counter_name = ?", (counter,)):
        k = name.split("(")[0].replace("void ", "").replace("hop::", "")
        k = k.split("<")[0]
        a = out.setdefault(k, [0, 0.0, 0.0])
        a[0] += 1
        a[1] += float(value)
        a[2] += float(dur)
    return out


def main():
    fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
    write = per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, [0, 0.0, 0.0])
        w = write.get(k, [0, 0.0, 0.0])
        n = max(f[0], w[0], 1)
        rd = f[1] * 1024 / max(f[0], 1)
        wr = w[1] * 1024 / max(w[0], 1)
        res[k] = {"launches": n, "fetch_bytes_per_launch_raw": rd, "fetch_bytes_per_launch_x2": 2 * rd,
                  "write_bytes_per_launch": wr, "hbm_bytes_per_launch": 2 * rd + wr,
                  "avg_us_under_pmc": (f[2] / max(f[0], 1)) / 1e3}
    json.dump(res, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
        print(f"{k:<24} launches {v['launches']:>4}  fetch(x2) {v['fetch_bytes_per_launch_x2']/1e6:>10.2f} MB  write {v['write_bytes_per_launch']/1e6:>9.2f} MB  "
              f"avg {v['avg_us_under_pmc']:>9.1f} us  -> {v['hbm_bytes_per_launch']/max(v['avg_us_under_pmc'],1e-9)/1e3:>8.1f} GB/s")


if __name__ == "__main__":
    main()
