"""Where the ICP restatement departs from what PCL is documented to do (PCL 1.9 is absent: "parity unpinned"), how much does
the result care?  Oracle only (CPU).  Deviations (SURVEY.md 8(a) notes):
  * minimiser: one Gauss-Newton step applied about the centroid of the matched points (the oracle and the GPU) vs the same
    step about the origin vs the non-linear point-to-plane problem minimised to convergence, which is what
    pcl::registration::TransformationEstimationPointToPlane's Levenberg-Marquardt returns -- computed two independent ways
    (damped Gauss-Newton on (t, quaternion) with a forward-difference Jacobian; Gauss-Newton steps about the centroid);
  * the surface-normal rejector keeps n_src . n_tgt >= cos (here) vs > cos (PCL);
  * the relative-MSE 1e-10 stop is kept (here) vs dropped (PCL overwrites it with its own default).
The as-shipped chain (generate -> cluster 30 deg / 15 mm -> ICP on <= 100 -> cluster 5 deg / 3 mm -> computeLCP argmax) runs
once per variant on the C1 frame (the reference's example/depth7.png hand region) and on 60 synthetic frames.

What holds, and is asserted:
  1. `>` vs `>=` and the relative stop change NOTHING: the same pose bit for bit on every frame;
  2. every minimiser leaves the selected pose's translation within 1 mm of the default's on every synthetic frame, every
     variant passes the authors' ADI < 5 mm on every frame, and the mean rotation error against the ground truth is the same
     within 0.5 degree: the deviation costs no accuracy;
  3. the SELECTED rotation is not stable to one degree under ANY change of the inner solver -- not even between the two
     computations of the same non-linear minimiser: the early |mse - mse_prev| < 1e-6 stop and the argmax over
     near-equivalent cluster heads amplify 1e-6 differences on this near-symmetric object.  So north_star's 1 mm / 1 degree
     can only be asked of two runs of the SAME arithmetic (GPU vs oracle: tests/test_gpu_parity.py), and no variant is a
     better stand-in for PCL than another; the one-step form stays (it is the cheapest on the GPU)."""
import math
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

VARIANTS = {
    "default: one GN step about the centroid, >=, rel. stop": dict(minimiser=0, strict_normal=False, relative_stop=True),
    "strict > in the normal rejector": dict(minimiser=0, strict_normal=True, relative_stop=True),
    "no relative-MSE stop": dict(minimiser=0, strict_normal=False, relative_stop=False),
    "GN step about the origin": dict(minimiser=1, strict_normal=False, relative_stop=True),
    "non-linear minimiser, LM form": dict(minimiser=2, strict_normal=True, relative_stop=False),
    "non-linear minimiser, inner GN steps": dict(minimiser=3, strict_normal=True, relative_stop=False),
}
IMMATERIAL = ("strict > in the normal rejector", "no relative-MSE stop")
SYM = [180, 180, 180]


def _rot_err_deg(Ra, Rb):
    """modulo the ellipsoid's 180 degree symmetries (object_symmetry.ellipse, config_autodataset.yaml)"""
    best = 180.0
    for F in ([1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1]):
        c = (np.trace(Ra.astype(np.float64).T @ (Rb.astype(np.float64) @ np.diag(F))) - 1) / 2
        best = min(best, math.degrees(math.acos(max(-1.0, min(1.0, float(c))))))
    return best


def _adi(model, est, gt):
    a = model @ est[:3, :3].T.astype(np.float64) + est[:3, 3]
    b = model @ gt[:3, :3].T.astype(np.float64) + gt[:3, 3]
    return float(cKDTree(a).query(b, k=1)[0].mean())


def _chain(orc, xyz, nrm, conf, mx5, mn5, mx1, mn1, keys):
    keep = conf >= 0.8
    S, Sn = xyz[keep], nrm[keep]
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(xyz, nrm, conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert len(ol)
    k1 = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, SYM)
    p1, l1 = op[k1][:100], ol[k1][:100]
    out = {}
    for name, v in VARIANTS.items():
        p2, it, cv = orc.icp_refine_batch_variant(S, Sn, mx5, mn5, p1, 10, 45.0, 0.01, **v)
        k2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, SYM)
        p3 = p2[k2]
        s3 = orc.compute_lcp_batch(S, Sn, mx1, mn1, p3, 0.001, 10.0)
        out[name] = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    return out


def test_variant_default_equals_the_oracle_entry_point(orc, hop):
    synth = hop.synth
    sc = synth.make_scene(1500, seed=1003)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, 12, seed=4, max_rot_deg=10.0, max_trans=0.005)
    a = orc.icp_refine_batch(sc.xyz, sc.nrm, mx5, mn5, poses, 10, 45.0, 0.01, use_tree=True)
    b = orc.icp_refine_batch_variant(sc.xyz, sc.nrm, mx5, mn5, poses, 10, 45.0, 0.01, 0, False, True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    # one ICP iteration from a 3 degree / 2 mm start: the two computations of the non-linear minimiser agree to 1e-4 (the rotation of this near-symmetric object is weakly determined), the
    # single Gauss-Newton step is within its second-order term of them
    start = synth.replay_poses(sc.gt_pose, 6, seed=9, max_rot_deg=3.0, max_trans=0.002)
    lm = orc.icp_refine_batch_variant(sc.xyz, sc.nrm, mx5, mn5, start, 1, 45.0, 0.01, 2, False, True)[0]
    gn3 = orc.icp_refine_batch_variant(sc.xyz, sc.nrm, mx5, mn5, start, 1, 45.0, 0.01, 3, False, True)[0]
    gn1 = orc.icp_refine_batch_variant(sc.xyz, sc.nrm, mx5, mn5, start, 1, 45.0, 0.01, 0, False, True)[0]
    assert np.abs(lm - gn3).max() < 1e-4
    assert np.abs(gn1 - gn3).max() < 2e-3 and np.abs(gn1 - gn3).max() > 1e-7


def test_icp_deviations_on_the_c1_frame(orc, hop, golden_dir):
    """example/depth7.png holds a hand and no ellipse: the chain's winner is one of many poor local optima, the least stable
    case there is.  The two immaterial deviations still change nothing."""
    synth = hop.synth
    g = np.load(os.path.join(golden_dir, "depth7_hand_region.npz"))
    xyz, nrm = g["xyz"], g["nrm"]
    out = _chain(orc, xyz, nrm, np.ones(len(xyz), np.float32), *synth.ellipsoid_model_spacing(0.005), *synth.ellipsoid_model(4000), synth.ppf_key_table())
    ref = out[next(iter(VARIANTS))]
    for name in IMMATERIAL:
        assert np.array_equal(out[name], ref), name


def test_icp_deviations_on_60_synthetic_frames(orc, hop):
    synth = hop.synth
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    model = mx1.astype(np.float64)
    rows = {name: [] for name in VARIANTS}
    for f in range(60):
        sc = synth.make_scene(1500, seed=1000 + f)
        out = _chain(orc, sc.xyz, sc.nrm, sc.conf, mx5, mn5, mx1, mn1, keys)
        ref = out[next(iter(VARIANTS))]
        for name, pose in out.items():
            rows[name].append((float(np.linalg.norm(pose[:3, 3] - ref[:3, 3])), _rot_err_deg(pose[:3, :3], ref[:3, :3]),
                               _rot_err_deg(pose[:3, :3], sc.gt_pose[:3, :3]), _adi(model, pose, sc.gt_pose), bool(np.array_equal(pose, ref))))
    stat = {name: np.array(v, dtype=np.float64) for name, v in rows.items()}
    for name, a in stat.items():
        print("%-56s vs default: max %.3f mm, %.2f deg (> 1 deg on %2d frames) | vs truth: mean rot %.2f deg, max ADI %.2f mm"
              % (name, 1e3 * a[:, 0].max(), a[:, 1].max(), int((a[:, 1] > 1).sum()), a[:, 2].mean(), 1e3 * a[:, 3].max()))
    base = stat[next(iter(VARIANTS))]
    for name in IMMATERIAL:                                   # 1.
        assert stat[name][:, 4].all(), name
    for name, a in stat.items():                              # 2.
        assert a[:, 0].max() < 1e-3, name
        assert a[:, 3].max() < 0.005, name                    # the authors' recall threshold (scripts/eval_all.py:77)
        assert abs(a[:, 2].mean() - base[:, 2].mean()) < 0.5, name
    # 3. the instability is not a property of the one-step form: the two computations of the same minimiser disagree too
    lm, gn3 = stat["non-linear minimiser, LM form"], stat["non-linear minimiser, inner GN steps"]
    assert (np.abs(lm[:, 1] - gn3[:, 1]) > 1.0).sum() >= 1
