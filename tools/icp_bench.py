#!/usr/bin/env python
"""Micro-benchmark of the ICP and computeLCP stages on the C2 hypothesis set (bench.py's sizes), one frame at a time:
generate once (2048 base trials), keep the 10 240 best, then time each ICP variant / computeLCP variant on the SAME
starting poses with HIP events (hop_timing_get) and compare the results between variants.

    python tools/icp_bench.py [--reps 3] [--scene 20000] [--hyps 10240]

Variants: ICP nn_mode 2 (split kernels), 3 (fused, chained increments), 4 (fused, composed increments), 6 (float moment sums), 7 (integer moment
sums on the matrix cores), 7dot2 (the same on the vector units); computeLCP nn_mode 2 (ordered sum over a term table), 3 (in-wave partial sums).
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--scene", type=int, default=20000)
    ap.add_argument("--model", type=int, default=5000)
    ap.add_argument("--bases", type=int, default=2048)
    ap.add_argument("--hyps", type=int, default=10240)
    ap.add_argument("--lcp-modes", default="2,3")
    ap.add_argument("--icp-modes", default="4,6,7,7dot2")
    args = ap.parse_args()
    import hop_loader
    hop = hop_loader.load()
    from hop_amd import api
    if os.environ.get("HOP_LIB"):
        api.LIB_PATH = os.environ["HOP_LIB"]  # an alternative build of libhop.so (experiments)
    synth = hop.synth
    sc = synth.make_scene(args.scene, seed=7)
    mx, mn = synth.ellipsoid_model(args.model)
    c = api.Context(0)
    c.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    c.set_model(api.HOP_MODEL_5MM, mx, mn)
    c.set_model(api.HOP_MODEL_1MM, mx, mn)
    c.set_ppf_keys(synth.ppf_key_table())
    o = c.default_s4pcs_opts(sample_size=100, success_quadrilaterals=args.bases, max_time_seconds=0, n_trials=args.bases, random_seed=5489)
    c.s4pcs_generate(o, download=False)
    c.hypos_keep_topk(args.hyps)
    poses0, scores0, _ = c.hypos_download()
    H = len(poses0)
    N = c.L.hop_scene_size(c.h)
    out = {"H": H, "N": N, "M": args.model, "icp": {}, "lcp": {}}
    bytes_per_hyp = 24 * (N + args.model) + 72
    res = {}
    for name in args.icp_modes.split(","):
        mode = int(name.replace("dot2", ""))
        os.environ["HOP_ICP_MFMA"] = "0" if name.endswith("dot2") else "1"   # "7dot2": nn_mode 7 with the moment sums on the vector units (k_icp_fusedq_momi)
        ms, nl, ms_all = [], [], []
        for rep in range(args.reps + 1):
            c.hypos_upload(poses0, scores0)
            c.timing_enable(True)
            c.timing_reset()
            it, cv = c.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
            c.synchronize()
            t = c.timing_get()
            c.timing_enable(False)
            if rep > 0:
                ms.append(t["ms_icp_nn"] + t["ms_icp_accum"])
                ms_all.append(t["ms_icp_nn"] + t["ms_icp_accum"] + t["ms_icp_solve"])  # nn_mode 5: the minimiser's passes are in the solve span
                nl.append(t["n_icp_nn_launches"])
        p, _, _ = c.hypos_download()
        res[name] = (it.copy(), cv.copy(), p.copy())
        hyp_iters = int(it.sum())
        m = float(np.mean(ms))
        out["icp"][name] = {"ms_nn_plus_accum": m, "ms_with_solve": float(np.mean(ms_all)), "launches": int(nl[0]), "hyp_iters": hyp_iters,
                            "us_per_launch": 1e3 * m / max(nl[0], 1),
                            "algorithmic_GBps": hyp_iters * bytes_per_hyp / (m * 1e-3) / 1e9,
                            "frac_of_8TBps": hyp_iters * bytes_per_hyp / (m * 1e-3) / 1e9 / 8000.0}
    names = list(res)
    for a in names[1:]:
        b = names[0]
        out["icp"][a]["vs_" + b] = {"iters_equal": float((res[a][0] == res[b][0]).mean()), "conv_equal": float((res[a][1] == res[b][1]).mean()),
                                   "max_pose_diff": float(np.abs(res[a][2] - res[b][2]).max()),
                                   "p99_pose_diff": float(np.quantile(np.abs(res[a][2] - res[b][2]).reshape(H, -1).max(axis=1), 0.99))}
    refined = res[names[-1]][2]
    lres = {}
    for name in args.lcp_modes.split(","):
        mode = int(name)
        ms = []
        for rep in range(args.reps + 1):
            c.hypos_upload(refined)
            c.timing_enable(True)
            c.timing_reset()
            try:
                best, score, idx = c.lcp_select_best(0.001, 10.0, mode)
            except Exception as e:
                out["lcp"][name] = {"error": str(e)}
                break
            c.synchronize()
            t = c.timing_get()
            c.timing_enable(False)
            if rep > 0:
                ms.append((t["ms_lcp_fwd"] + t["ms_lcp_rev"], t["ms_lcp_sum"]))
        else:
            _, s, _ = c.hypos_download()
            lres[name] = (s.copy(), idx, score)
            a = np.mean(ms, axis=0)
            out["lcp"][name] = {"ms_cells": float(a[0]), "ms_sum": float(a[1]), "best_index": idx, "best_score": score,
                                "algorithmic_GBps": H * bytes_per_hyp / ((a[0] + a[1]) * 1e-3) / 1e9}
    ln = list(lres)
    for a in ln[1:]:
        b = ln[0]
        d = np.abs(lres[a][0] - lres[b][0]) / np.maximum(np.abs(lres[b][0]), 1e-6)
        out["lcp"][a]["vs_" + b] = {"max_rel_diff": float(d[lres[b][0] > 1.0].max()) if (lres[b][0] > 1.0).any() else 0.0,
                                   "same_best": bool(lres[a][1] == lres[b][1])}
    print(json.dumps(out))
    c.close()


if __name__ == "__main__":
    main()
