"""Row S2 with the reference's own minimiser (hop_icp_refine nn_mode 5 and 6, csrc/hop_icp_lm.hip) -- `pytest -m gpu` on an MI355X.

The reference's ICP (Utils::runICP, Utils.cpp:188-229) minimises point-to-plane with PCL's TransformationEstimationPointToPlane =
Eigen::LevenbergMarquardt<NumericalDiff<..>, float>.  Eigen's code is vendored in the reference and compiled in place
(oracle/ref_icp_driver.cpp -> tests/golden/icp_lm_*.npz).  Here:
  * GPU vs the CPU restatement of the same algorithm (oracle lm_*, itself pinned to the goldens by tests/test_icp_lm_oracle.py):
    iteration counts equal and poses equal to float rounding (in fact bit for bit) on >= 93 % of the hypotheses.  The rest: the
    GPU adds the 28 sums of a pass in Morton order through a reduction tree, the oracle in source order; the 1e-16 difference,
    amplified by the conditioning of the ellipse's slides, now and then flips the last bit of a step when it is rounded to float,
    and the forward-difference Jacobian at |x| ~ 1e-3 (10 % noise per entry, by construction of PCL's float NumericalDiff) turns
    that bit into 1e-3 of pose -- the same sensitivity Eigen's own run has to its build flags (next tests);
  * GPU vs the golden vectors of Eigen's own run (C1 = example/depth7.png hand region, and a C2-style subset): as close to the
    reference's default build as the reference's -march=native build is (the fixtures hold both).
"""
import os

import numpy as np
import pytest

from test_icp_lm_oracle import assert_as_close_as_the_other_build, c1_inputs, c2sub_inputs, closeness, pose_deltas

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(hop):
    from hop_amd import api as _api
    _api.lib()
    return _api


@pytest.fixture()
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


def _gpu_lm(ctx, api, xyz, nrm, conf, mx5, mn5, poses, max_hypotheses=0, nn_mode=5):
    ctx.set_scene(xyz, nrm, conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.hypos_upload(poses)
    it, cv = ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=max_hypotheses, nn_mode=nn_mode, want_stats=True)
    p, _, _ = ctx.hypos_download()
    return p, it, cv


@pytest.mark.parametrize("ns,nh,rot,trans,seed", [(1500, 64, 10.0, 0.005, 1003), (4000, 128, 25.0, 0.012, 7), (20000, 256, 30.0, 0.015, 7)])
def test_lm_mode_equals_the_restated_minimiser(ctx, api, orc, hop, ns, nh, rot, trans, seed):
    synth = hop.synth
    sc = synth.make_scene(ns, seed=seed)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, nh, seed=4, max_rot_deg=rot, max_trans=trans)
    keep = sc.conf >= 0.8
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx5, mn5, poses, 10, 45.0, 0.01)
    pg, itg, cvg = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    assert np.array_equal(cvg, cvo)
    same = itg == ito
    d = np.abs(pg.reshape(-1, 16) - po.reshape(-1, 16)).max(1)
    tight = same & (d <= 2e-5)
    assert tight.sum() >= int(0.93 * nh), (int(same.sum()), int(tight.sum()), float(d.max()))
    t, r = pose_deltas(pg, po)
    assert ((t < 1) & (r < 1)).sum() >= int(0.98 * nh)


@pytest.mark.parametrize("ns,nh,rot,trans,seed", [(1500, 64, 10.0, 0.005, 1003), (4000, 128, 25.0, 0.012, 7), (20000, 256, 30.0, 0.015, 7)])
def test_moment_mode_equals_the_minimiser_in_exact_arithmetic(ctx, api, orc, hop, ns, nh, rot, trans, seed):
    """nn_mode 6: Eigen's minimiser evaluated from the 13 x 13 moment matrix of the correspondences (the residual is linear in
    [R | t]), one pass per ICP iteration -- against the oracle's per-point statement of the same thing (lm_pass_exact: the
    reference's algorithm with every residual in double instead of float)."""
    synth = hop.synth
    sc = synth.make_scene(ns, seed=seed)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, nh, seed=4, max_rot_deg=rot, max_trans=trans)
    keep = sc.conf >= 0.8
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx5, mn5, poses, 10, 45.0, 0.01, exact=True)
    pg, itg, cvg = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses, nn_mode=6)
    assert np.array_equal(cvg, cvo)
    assert (itg == ito).sum() >= int(0.97 * nh)
    d = np.abs(pg.reshape(-1, 16) - po.reshape(-1, 16)).max(1)
    assert np.median(d) < 2e-6 and np.percentile(d, 90) < 5e-5, (float(np.median(d)), float(np.percentile(d, 90)))
    t, r = pose_deltas(pg, po)
    assert ((t < 1) & (r < 1)).sum() >= int(0.98 * nh)


def test_moment_mode_vs_eigens_own_run(ctx, api, hop, golden_dir):
    """... and against the golden vectors of Eigen's float run: closer to the reference's default build than (or as close as) the
    reference's -march=native build is -- C2-style subset and the C1 frame (example/depth7.png)."""
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    sc = hop.synth.make_scene(4000, seed=7)
    mx5, mn5 = hop.synth.ellipsoid_model_spacing(0.005)
    p, it, cv = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, g["poses_in"], nn_mode=6)
    mine, native = assert_as_close_as_the_other_build(p, g, slack=3)
    assert mine[0] >= 88 and (it == g["iterations"]).sum() >= 88 and np.array_equal(cv, g["converged"])
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    xyz, nrm, mx5, mn5 = c1_inputs(hop, golden_dir)
    p, it, cv = _gpu_lm(ctx, api, xyz, nrm, np.ones(len(xyz), np.float32), mx5, mn5, g["poses_in"], max_hypotheses=100, nn_mode=6)
    assert_as_close_as_the_other_build(p, g, slack=5)
    assert (cv == g["converged"]).sum() >= 88


def test_moment_mode_falls_back_to_the_per_evaluation_form_without_packed_lists(ctx, api, orc, hop):
    """a model of >= 65535 points has no packed lists: nn_mode 6 runs the same minimiser through nn_mode 5's passes"""
    synth = hop.synth
    sc = synth.make_scene(800, seed=11)
    mx, mn = synth.ellipsoid_model(70000)
    poses = synth.replay_poses(sc.gt_pose, 6, seed=3, max_rot_deg=4.0, max_trans=0.002)
    pa, ita, cva = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx, mn, poses, nn_mode=6)
    pb, itb, cvb = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx, mn, poses, nn_mode=5)
    assert np.array_equal(pa, pb) and np.array_equal(ita, itb) and cva.all()


def test_lm_mode_few_points_and_not_converged(ctx, api, orc, hop):
    """hypotheses far from the scene (no correspondences -> not converged -> identity, Utils.cpp:218-225) next to good ones"""
    synth = hop.synth
    sc = synth.make_scene(600, seed=5)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, 12, seed=2, max_rot_deg=5.0, max_trans=0.003)
    poses[3, :3, 3] += 0.5
    poses[7, :3, 3] -= 0.3
    keep = sc.conf >= 0.8
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx5, mn5, poses, 10, 45.0, 0.01)
    pg, itg, cvg = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    assert cvo[3] == 0 and cvo[7] == 0 and np.array_equal(cvg, cvo) and np.array_equal(itg, ito)
    assert np.array_equal(pg[3], poses[3]) and np.array_equal(pg[7], poses[7])
    assert np.abs(pg - po).max() < 2e-5


def test_lm_mode_vs_eigens_own_run_c2sub(ctx, api, hop, golden_dir):
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    sc = hop.synth.make_scene(4000, seed=7)
    mx5, mn5 = hop.synth.ellipsoid_model_spacing(0.005)
    p, it, cv = _gpu_lm(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, g["poses_in"])
    mine, native = assert_as_close_as_the_other_build(p, g, slack=3)
    assert mine[0] >= 88 and (it == g["iterations"]).sum() >= 88 and np.array_equal(cv, g["converged"])
    # and closer to the reference than the one-step Gauss-Newton modes are
    ctx.hypos_upload(g["poses_in"])
    ctx.icp_refine(10, 45.0, 0.01, nn_mode=3)
    p3, _, _ = ctx.hypos_download()
    assert closeness(p3, g["poses_out"])[0] < mine[0]


def test_lm_mode_vs_eigens_own_run_c1_depth7(ctx, api, hop, golden_dir):
    """BASELINE configs[0]: the hand region of the reference's example/depth7.png, refineByICP's <= 100 hypotheses"""
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    xyz, nrm, mx5, mn5 = c1_inputs(hop, golden_dir)
    p, it, cv = _gpu_lm(ctx, api, xyz, nrm, np.ones(len(xyz), np.float32), mx5, mn5, g["poses_in"], max_hypotheses=100)
    assert_as_close_as_the_other_build(p, g, slack=4)
    assert (cv == g["converged"]).sum() >= 90
