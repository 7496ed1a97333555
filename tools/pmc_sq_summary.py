#!/usr/bin/env python
"""Per-kernel SQ / TCP / TCC counter summary from rocprofv3 PMC passes (one results database per pass; the passes are
collected separately with --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes).

    python tools/pmc_sq_summary.py OUT.txt "command" PASS1.db [PASS2.db ...]

Per kernel (summed over its dispatches, then divided by the launches): every counter found, and the derived figures
the DESIGN text quotes:
  valu_issue_frac   = SQ_ACTIVE_INST_VALU * 4 / (SQ_BUSY_CYCLES-like denominator)   (see below)
  insts_per_wave    = SQ_INSTS_VALU / SQ_WAVES  (when SQ_WAVES was collected)
  l1_hit            = 1 - TCP_TCC_READ_REQ / TCP_TOTAL_CACHE_ACCESSES
  l2_hit            = TCC_HIT / (TCC_HIT + TCC_MISS)
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (same guide).  The share of a wave's
resident time spent issuing VALU is SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES when both were collected in the same pass
(different passes see different dispatch timings, the ratio is then approximate).
"""
import re
import sqlite3
import sys


def kname(name):
    k = name.split("(")[0].replace("void ", "").replace("hop::", "")
    k = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::trampoline_kernel.*", "rocprim", k)
    return k.split("<")[0]


def main():
    out, cmd, dbs = sys.argv[1], sys.argv[2], sys.argv[3:]
    per = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        seen = {}
        for name, cname, value, dur in cur.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            k = kname(name)
            e = per.setdefault(k, {})
            c = e.setdefault(cname, [0, 0.0, 0.0])
            c[0] += 1
            c[1] += float(value)
            c[2] += float(dur)
    derived = {}
    lines = [f"# rocprofv3 --kernel-trace --pmc ... per-kernel counter summary (values per launch; avg_us under PMC)", f"# command: {cmd}"]
    order = sorted(per, key=lambda k: -max(c[2] for c in per[k].values()))
    for k in order:
        e = per[k]
        n = max(c[0] for c in e.values())
        avg_us = max(c[2] / max(c[0], 1) for c in e.values()) / 1e3
        v = {cn: c[1] / max(c[0], 1) for cn, c in e.items()}
        lines.append(f"{k}: launches {n}, avg {avg_us:.1f} us")
        for cn in sorted(v):
            lines.append(f"    {cn:<32}{v[cn]:>18.0f}")
        d = []
        if "SQ_ACTIVE_INST_VALU" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            d.append(f"VALU-issue share of wave-resident time {v['SQ_ACTIVE_INST_VALU'] / v['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_ACTIVE_INST_ANY" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            d.append(f"any-issue share {v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            d.append(f"parked (s_waitcnt) share {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
        if "SQ_WAIT_INST_ANY" in v and "SQ_WAVE_CYCLES" in v and v["SQ_WAVE_CYCLES"] > 0:
            d.append(f"issue-stall share {v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES']:.3f}")
        dj = derived.setdefault(k, {"launches": n, "avg_us_under_pmc": avg_us})
        if "SQ_ACTIVE_INST_VALU" in v and avg_us > 0:
            dj["valu_busy"] = v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * avg_us * 1e-6 * 2.4e9)
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v and v["SQ_WAVES"] > 0:
            dj["valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in v and avg_us > 0:
            dj["l1_accesses_per_clk_cu"] = v["TCP_TOTAL_CACHE_ACCESSES_sum"] / (256 * avg_us * 1e-6 * 2.4e9)
            if "TCP_TCC_READ_REQ_sum" in v and v["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
                dj["l1_hit"] = 1 - v["TCP_TCC_READ_REQ_sum"] / v["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
            dj["l2_hit"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        if "SQ_ACTIVE_INST_VALU" in v and avg_us > 0:
            # quad-cycles of VALU issue, 1024 SIMDs; clock taken as 2.4 GHz
            d.append(f"SIMD VALU busy {v['SQ_ACTIVE_INST_VALU'] * 4 / (1024 * avg_us * 1e-6 * 2.4e9):.3f} (4 cyc/quad, 1024 SIMDs, 2.4 GHz)")
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v and v["SQ_WAVES"] > 0:
            d.append(f"VALU insts per wave {v['SQ_INSTS_VALU'] / v['SQ_WAVES']:.0f}")
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in v and "TCP_TCC_READ_REQ_sum" in v and v["TCP_TOTAL_CACHE_ACCESSES_sum"] > 0:
            d.append(f"L1 hit {1 - v['TCP_TCC_READ_REQ_sum'] / v['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
            d.append(f"L1 accesses/clk/CU {v['TCP_TOTAL_CACHE_ACCESSES_sum'] / (256 * avg_us * 1e-6 * 2.4e9):.2f}")
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v and v["TCC_HIT_sum"] + v["TCC_MISS_sum"] > 0:
            d.append(f"L2 hit {v['TCC_HIT_sum'] / (v['TCC_HIT_sum'] + v['TCC_MISS_sum']):.3f}")
        for x in d:
            lines.append("    -> " + x)
    open(out, "w").write("\n".join(lines) + "\n")
    # the derived figures per kernel as JSON next to the text (bench.py puts them into the line's roofline object: profiles/pmc_sq_latest.json)
    import json
    import os
    json.dump({"_source": cmd, **derived}, open(os.path.splitext(out)[0] + ".json", "w"), indent=1)
    print("\n".join(lines[:120]))


if __name__ == "__main__":
    main()
