#!/usr/bin/env python
"""The experiment behind the 12-bit grid of nn_mode 7's moment form (DESIGN.md section 4 "ICP, nn_mode 7"; VERDICT r04 "has no committed artifact").

For grids of 9 .. 20 bits (|U| <= 2^bits) the oracle's minimiser 7 refines the hypotheses of the as-shipped chain (generate -> cluster -> the
first 100) on a few frames; every width is compared
  * with the EXACT-arithmetic form of the same minimiser (no grid: what the grid approximates), and
  * with Eigen::LevenbergMarquardt itself run from the reference's vendored sources (oracle/_ref/libref_icp.so), next to which the
    reference's OTHER build (-march=native) is the yardstick for "the noise the reference's own float run carries".
CPU only (the oracle is test infrastructure; the GPU's nn_mode 7 returns the bits of the 12-bit row).

    python tools/mom_grid_widths.py --out profiles/r05_mom_grid_widths.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from icp_lm_deltas import SYM, rot_deg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--scene", type=int, default=1500)
    ap.add_argument("--bits", default="9,10,11,12,13,14,16,18,20")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    import hop_loader
    import orc
    hop = hop_loader.load()
    synth = hop.synth
    orc.build()
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    keys = synth.ppf_key_table()
    widths = [int(b) for b in args.bits.split(",")]
    rows = {("moment_%d_bits" % b): {"dt_exact": [], "dr_exact": [], "dt_ref": [], "dr_ref": [], "it_exact": 0, "it_ref": 0} for b in widths}
    rows["exact_form"] = {"dt_ref": [], "dr_ref": [], "it_ref": 0}
    rows["ref_native_build"] = {"dt_ref": [], "dr_ref": [], "it_ref": 0}
    n_h = 0
    frames = []
    g = np.load(os.path.join(ROOT, "tests", "golden", "depth7_hand_region.npz"))
    frames.append(("c1_depth7", g["xyz"], g["nrm"], np.ones(len(g["xyz"]), np.float32)))
    for f in range(args.frames - 1):
        sc = synth.make_scene(args.scene, seed=1000 + f)
        frames.append(("synthetic_%d" % (1000 + f), sc.xyz, sc.nrm, sc.conf))
    for name, xyz, nrm, conf in frames:
        keep = conf >= 0.8
        S, Sn = xyz[keep], nrm[keep]
        oo = orc.OracleS4PCS()
        oo.set_keys(keys)
        oo.run(xyz, nrm, conf, mx5, mn5, 1)
        op, ol = oo.hypos()
        k1 = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, SYM)
        p1 = np.ascontiguousarray(op[k1][:100])
        n_h += len(p1)
        orc.ref_icp_use(native=False)
        pr, itr, _ = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, p1, 10, 45.0, 0.01, ref=True)
        orc.ref_icp_use(native=True)
        pn, itn, _ = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, p1, 10, 45.0, 0.01, ref=True)
        orc.ref_icp_use(native=False)
        pe, ite, _ = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, p1, 10, 45.0, 0.01, exact=True)

        def add(row, P, it, vs_exact):
            for h in range(len(p1)):
                row["dt_ref"].append(1e3 * float(np.linalg.norm(P[h][:3, 3] - pr[h][:3, 3])))
                row["dr_ref"].append(rot_deg(P[h][:3, :3], pr[h][:3, :3]))
                if vs_exact:
                    row["dt_exact"].append(1e3 * float(np.linalg.norm(P[h][:3, 3] - pe[h][:3, 3])))
                    row["dr_exact"].append(rot_deg(P[h][:3, :3], pe[h][:3, :3]))
            row["it_ref"] += int(np.sum(it == itr))
            if vs_exact:
                row["it_exact"] += int(np.sum(it == ite))
        add(rows["exact_form"], pe, ite, False)
        add(rows["ref_native_build"], pn, itn, False)
        for b in widths:
            orc.set_mom_bits(b)
            pm, itm, _ = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, p1, 10, 45.0, 0.01, moment=True)
            add(rows["moment_%d_bits" % b], pm, itm, True)
        orc.set_mom_bits(orc.MOM_BITS)
        print(name, len(p1), "hypotheses", flush=True)
    res = {"what": "nn_mode 7's moment form (oracle minimiser 7) at grids of 2^bits against the exact-arithmetic form of the same minimiser and against "
                   "Eigen::LevenbergMarquardt run from the reference's vendored sources; as-shipped chain (generate, cluster 30 deg / 15 mm, first 100), CPU oracle",
           "frames": [f[0] for f in frames], "hypotheses": n_h, "shipped_bits": orc.MOM_BITS, "rows": {}}
    for k, r in rows.items():
        t, d = np.array(r["dt_ref"]), np.array(r["dr_ref"])
        o = {"vs_eigen_run": {"within_1mm_1deg": int(((t < 1) & (d < 1)).sum()), "iterations_equal": r["it_ref"], "dt_mm_median": float(np.median(t)), "drot_deg_median": float(np.median(d))}}
        if "dt_exact" in r:
            t, d = np.array(r["dt_exact"]), np.array(r["dr_exact"])
            o["vs_exact_form"] = {"within_1mm_1deg": int(((t < 1) & (d < 1)).sum()), "iterations_equal": r["it_exact"], "dt_um_median": 1e3 * float(np.median(t)),
                                  "dt_um_p95": 1e3 * float(np.percentile(t, 95)), "drot_deg_median": float(np.median(d)), "drot_deg_p95": float(np.percentile(d, 95))}
        res["rows"][k] = o
    s = json.dumps(res, indent=1)
    print(s)
    if args.out:
        open(args.out, "w").write(s + "\n")


if __name__ == "__main__":
    main()
