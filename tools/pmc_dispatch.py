"""Per-dispatch counters of the kernels whose name contains a pattern, from a rocprofv3 --kernel-trace --pmc database:
duration, waves, VALU / SALU / VMEM instructions per wave, s_waitcnt share.

    rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY -d /tmp/p -o b -- <cmd>
    python tools/pmc_dispatch.py /tmp/p/*/b_results.db "k_cell_list_local<"
"""
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
pat = sys.argv[2]
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
rows = list(cur.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size from counters_collection")) if 'grid_size' in cols else [(a,b,c,d,e,0) for a,b,c,d,e in cur.execute("select dispatch_id, kernel_name, counter_name, value, duration from counters_collection")]
d = collections.OrderedDict()
for did, kn, cn, v, dur, gs in rows:
    if pat not in kn: continue
    e = d.setdefault(did, {"name": kn.split('(')[0][-40:], "dur": dur, "grid": gs})
    e[cn] = e.get(cn, 0) + v
for did, e in list(d.items())[-12:]:
    w = e.get("SQ_WAVES", 0)
    print(e["name"], "grid", e["grid"], "dur_us %.1f" % (e["dur"] / 1e3), "waves", int(w), "valu/wave %.0f" % (e.get("SQ_INSTS_VALU", 0) / max(w, 1)), "salu/wave %.0f" % (e.get("SQ_INSTS_SALU", 0) / max(w, 1)), "vmem/wave %.0f" % (e.get("SQ_INSTS_VMEM_RD", 0) / max(w, 1)), "waitshare %.2f" % (e.get("SQ_WAIT_ANY", 0) / max(e.get("SQ_WAVE_CYCLES", 1), 1)))
