#!/usr/bin/env python
"""tools/icp_lm_deltas.py -- TEST INFRASTRUCTURE (uses oracle/): how far apart are the ICP minimisers?

refineByICP's arithmetic is PCL's TransformationEstimationPointToPlane = Eigen::LevenbergMarquardt on a forward-difference
float Jacobian (Utils.cpp:200-216).  Eigen's code is in the reference tree and is compiled in place (oracle/_ref/libref_icp.so).
This tool runs the as-shipped chain (generate -> cluster 30 deg / 15 mm -> ICP on <= 100 -> cluster 5 deg / 3 mm -> computeLCP
arg-max, main_realdata_auto.cpp:187-204) on the C1 frame (example/depth7.png hand region) and on 60 synthetic frames with the
ICP minimiser switched, everything else identical (CPU oracle):
    ref        Eigen's LevenbergMarquardt, default g++ build (SSE2, no contraction)  -- the golden vectors' build
    ref_native the same source, -march=native (wider packets, FMA): a second BUILD of the reference
    gpu_*      with --gpu: hop_icp_refine on the MI355X, nn_mode 5 (the restated minimiser in HIP, float-faithful), nn_mode 6 (the same from moment sums, one pass per ICP iteration) and nn_mode 3 (one GN step)
    lm         the restatement of that algorithm (oracle/hop_oracle.cpp lm_*; what the GPU's nn_mode 5 computes)
    lm_exact   the same minimiser with every residual in exact (double) arithmetic instead of float -- what the GPU's nn_mode 6 computes
               from the 13 x 13 moment matrix of the correspondences, one pass per ICP iteration
    lm_moment  the moment form with integer-exact sums on a 12-bit grid and an IEEE-only solve (oracle minimiser 7) -- what the GPU's
               nn_mode 7 returns BIT FOR BIT (the mode the mirrors and the bench run)
    gn         one Gauss-Newton step about the matched centroid per ICP iteration (nn_mode 0-4)
and reports, against `ref`: per refined hypothesis (all <= 100 per frame) and for the SELECTED pose, translation / rotation
differences.  `ref_native` vs `ref` is the reference's own build-to-build spread: no implementation can be asked to be closer to
the reference than the reference is to itself.

    python tools/icp_lm_deltas.py --frames 60 --out profiles/r03_icp_lm_deltas.json
"""
import argparse
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
SYM = [180, 180, 180]


def rot_deg(Ra, Rb, sym=False):
    best = 180.0
    for F in ([1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]) if sym else ([1, 1, 1],):
        c = (np.trace(Ra.astype(np.float64).T @ (Rb.astype(np.float64) @ np.diag(F))) - 1) / 2
        best = min(best, math.degrees(math.acos(max(-1.0, min(1.0, float(c))))))
    return best


def chain(orc, xyz, nrm, conf, mx5, mn5, mx1, mn1, keys, variants):
    keep = conf >= 0.8
    S, Sn = xyz[keep], nrm[keep]
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(xyz, nrm, conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    k1 = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, SYM)
    p1, l1 = op[k1][:100], ol[k1][:100]
    out = {}
    for name, fn in variants.items():
        p2, it, cv = fn(S, Sn, mx5, mn5, p1)
        k2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, SYM)
        p3 = p2[k2]
        s3 = orc.compute_lcp_batch(S, Sn, mx1, mn1, p3, 0.001, 10.0)
        out[name] = dict(refined=p2, iters=it, conv=cv, selected=p3[int(np.flatnonzero(s3 == s3.max())[0])], n_clusters=len(k2))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=60)
    ap.add_argument("--scene", type=int, default=1500)
    ap.add_argument("--out", default="")
    ap.add_argument("--gpu", action="store_true", help="add hop_icp_refine on cuda:0 (nn_mode 5 = the minimiser above, nn_mode 3 = one Gauss-Newton step) as variants")
    args = ap.parse_args()
    import hop_loader
    import orc
    hop = hop_loader.load()
    synth = hop.synth
    orc.build()
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()

    def v_ref(native):
        def f(S, Sn, M, Mn, P):
            orc.ref_icp_use(native=native)
            return orc.icp_refine_batch_lm(S, Sn, M, Mn, P, 10, 45.0, 0.01, ref=True)
        return f
    variants = {
        "ref": v_ref(False),
        "ref_native": v_ref(True),
        "lm": lambda S, Sn, M, Mn, P: orc.icp_refine_batch_lm(S, Sn, M, Mn, P, 10, 45.0, 0.01, ref=False),
        "lm_exact": lambda S, Sn, M, Mn, P: orc.icp_refine_batch_lm(S, Sn, M, Mn, P, 10, 45.0, 0.01, exact=True),
        "lm_moment": lambda S, Sn, M, Mn, P: orc.icp_refine_batch_lm(S, Sn, M, Mn, P, 10, 45.0, 0.01, moment=True),
        "gn": lambda S, Sn, M, Mn, P: orc.icp_refine_batch(S, Sn, M, Mn, P, 10, 45.0, 0.01),
    }
    if args.gpu:
        from hop_amd import api
        ctx = api.Context(0)
        ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)

        def v_gpu(mode):
            def f(S, Sn, M, Mn, P):
                ctx.set_scene(S, Sn, np.ones(len(S), np.float32), 0.8)
                ctx.hypos_upload(P)
                it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
                return ctx.hypos_download()[0], it, cv
            return f
        variants["gpu_lm_nn_mode5"] = v_gpu(5)
        variants["gpu_lm_moments_nn_mode6"] = v_gpu(6)
        variants["gpu_lm_integer_moments_nn_mode7"] = v_gpu(7)
        variants["gpu_gn_nn_mode3"] = v_gpu(3)
    frames = []
    g = np.load(os.path.join(ROOT, "tests", "golden", "depth7_hand_region.npz"))
    frames.append(("c1_depth7", g["xyz"], g["nrm"], np.ones(len(g["xyz"]), np.float32), None))
    for f in range(args.frames):
        sc = synth.make_scene(args.scene, seed=1000 + f)
        frames.append(("synthetic_%d" % (1000 + f), sc.xyz, sc.nrm, sc.conf, sc.gt_pose))
    per_hyp = {k: {"dt_mm": [], "drot_deg": [], "iters_equal": 0, "n": 0} for k in variants if k != "ref"}
    sel = {k: [] for k in variants if k != "ref"}
    gt_rows = {k: [] for k in variants}
    for name, xyz, nrm, conf, gt in frames:
        out = chain(orc, xyz, nrm, conf, mx5, mn5, mx1, mn1, keys, variants)
        ref = out["ref"]
        for k, o in out.items():
            if gt is not None:
                gt_rows[k].append((1e3 * float(np.linalg.norm(o["selected"][:3, 3] - gt[:3, 3])), rot_deg(o["selected"][:3, :3], gt[:3, :3], sym=True)))
            if k == "ref":
                continue
            for a, b, ia, ib in zip(o["refined"], ref["refined"], o["iters"], ref["iters"]):
                per_hyp[k]["dt_mm"].append(1e3 * float(np.linalg.norm(a[:3, 3] - b[:3, 3])))
                per_hyp[k]["drot_deg"].append(rot_deg(a[:3, :3], b[:3, :3]))
                per_hyp[k]["iters_equal"] += int(ia == ib)
                per_hyp[k]["n"] += 1
            sel[k].append((name, 1e3 * float(np.linalg.norm(o["selected"][:3, 3] - ref["selected"][:3, 3])),
                           rot_deg(o["selected"][:3, :3], ref["selected"][:3, :3], sym=True)))
        print(name, {k: "%.3f mm %.3f deg" % (sel[k][-1][1], sel[k][-1][2]) for k in sel}, flush=True)
    res = {"what": "ICP minimisers against Eigen::LevenbergMarquardt from the reference's vendored Eigen (default g++ build); as-shipped chain, CPU oracle",
           "frames": [f[0] for f in frames][:2] + ["..."], "n_frames": len(frames), "per_refined_hypothesis": {}, "selected_pose": {}, "selected_vs_ground_truth": {}}
    for k, d in per_hyp.items():
        t, r = np.array(d["dt_mm"]), np.array(d["drot_deg"])
        res["per_refined_hypothesis"][k + "_vs_ref"] = {
            "n": d["n"], "iterations_equal": d["iters_equal"], "dt_mm_median": float(np.median(t)), "dt_mm_p95": float(np.percentile(t, 95)), "dt_mm_max": float(t.max()),
            "drot_deg_median": float(np.median(r)), "drot_deg_p95": float(np.percentile(r, 95)), "drot_deg_max": float(r.max()),
            "within_1mm_1deg": int(((t < 1) & (r < 1)).sum())}
    for k, rows in sel.items():
        t, r = np.array([x[1] for x in rows]), np.array([x[2] for x in rows])
        res["selected_pose"][k + "_vs_ref"] = {"frames": len(rows), "within_1mm_1deg": int(((t < 1) & (r < 1)).sum()), "dt_mm_max": float(t.max()), "drot_deg_max": float(r.max()),
                                                "c1_depth7": {"dt_mm": rows[0][1], "drot_deg": rows[0][2]},
                                                "frames_off": [x[0] for x in rows if not (x[1] < 1 and x[2] < 1)]}
    for k, rows in gt_rows.items():
        a = np.array(rows)
        res["selected_vs_ground_truth"][k] = {"dt_mm_mean": float(a[:, 0].mean()), "dt_mm_max": float(a[:, 0].max()), "drot_deg_mean": float(a[:, 1].mean()), "drot_deg_max": float(a[:, 1].max())}
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
