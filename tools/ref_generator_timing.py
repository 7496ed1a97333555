#!/usr/bin/env python
"""tools/ref_generator_timing.py -- TEST INFRASTRUCTURE, build container only (needs /root/reference through oracle/_ref).

BASELINE.md section 3 / SURVEY.md 8(d) "CPU baseline beside it": the REFERENCE's own generator (the OpenGR fork compiled in place,
-O2 build oracle/_ref/libref_s4pcs_o2.so, see oracle/ref_driver.cpp) timed at C2's sizes (20 000-point scene, 5 000-point model,
100 samples, delta 3 mm) next to the port (the oracle's generator, what bench.py's cpu_baseline times on the GPU box), same inputs,
one thread, one warm-up, median of 5.  The reference's loop runs at most 30 base trials per ComputeTransformation call
(congruentSetExplorationBase.hpp:90-100), so a run is `calls` calls on one matcher.

    python tools/ref_generator_timing.py --calls 2 --out profiles/r03_ref_generator_c2.json
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--calls", type=int, default=2)
    ap.add_argument("--scene", type=int, default=20000)
    ap.add_argument("--model", type=int, default=5000)
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    os.environ["OMP_NUM_THREADS"] = "1"
    import hop_loader
    import orc
    hop = hop_loader.load()
    synth = hop.synth
    orc.build()
    o2 = os.path.join(ROOT, "oracle", "_ref", "libref_s4pcs_o2.so")
    assert os.path.exists(o2), "make -C oracle ref"
    orc.REF_SO = o2          # the -O2 timing build behind the same wrapper
    orc._ref = None
    sc = synth.make_scene(args.scene, seed=7)
    mx, mn = synth.ellipsoid_model(args.model)
    keys = synth.ppf_key_table()
    keep = sc.conf >= 0.8
    P, Pn, Pc = sc.xyz[keep], sc.nrm[keep], sc.conf[keep]

    def run_ref():
        r = orc.RefS4PCS(success_quadrilaterals=10 ** 6, record_pairs=False, plain=True)
        r.set_keys(keys)
        t0 = time.perf_counter()
        n = r.run(P, Pn, Pc, mx, mn, args.calls)
        return time.perf_counter() - t0, n

    def run_port():
        oo = orc.OracleS4PCS(success_quadrilaterals=10 ** 6)
        oo.set_keys(keys)
        t0 = time.perf_counter()
        n = oo.run(P, Pn, Pc, mx, mn, args.calls)
        return time.perf_counter() - t0, n

    res = {}
    for name, fn in (("reference_opengr_O2", run_ref), ("port_oracle", run_port)):
        fn()
        ts, n = [], 0
        for _ in range(5):
            t, n = fn()
            ts.append(t)
        res[name] = {"hypotheses": int(n), "median_s": float(np.median(ts)), "all_s": ts, "R_gen_hyp_per_s": n / float(np.median(ts))}
        print(name, res[name], flush=True)
    model = "unknown"
    for ln in open("/proc/cpuinfo"):
        if ln.startswith("model name"):
            model = ln.split(":", 1)[1].strip()
            break
    out = {"what": "the reference's own generator (OpenGR fork, g++ -O2, one thread) and the port on the same C2-size inputs; build container",
           "cpu_model": model, "threads": 1, "scene_points": int(len(P)), "model_points": int(len(mx)), "sample_size": 100, "delta": 0.003,
           "ComputeTransformation_calls": args.calls, "base_trials": 30 * args.calls, "protocol": "one warm-up, median of 5",
           "same_hypothesis_count": res["reference_opengr_O2"]["hypotheses"] == res["port_oracle"]["hypotheses"], **res}
    s = json.dumps(out, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
