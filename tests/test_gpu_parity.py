"""GPU parity tests: the HIP path (through the C-ABI, libhop.so) against the CPU oracle and the golden vectors
emitted by the reference's own code.  Run with `pytest -m gpu` on an MI355X."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api(hop):
    from hop_amd import api as _api
    _api.lib()  # raises if libhop.so is missing: no fallback
    return _api


@pytest.fixture(scope="module")
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def synth(hop):
    return hop.synth


def _canon(pose, lcp):
    flat = pose.reshape(len(pose), 16)
    rot = flat[:, [0, 1, 2, 4, 5, 6, 8, 9, 10]]
    order = np.lexsort(tuple(rot[:, ::-1].T) + (-lcp,))
    return pose[order], lcp[order]


def _rot_err_deg(Ra, Rb):
    c = (np.trace(Ra.T @ Rb) - 1) / 2
    return math.degrees(math.acos(max(-1.0, min(1.0, float(c)))))


# ------------------------------------------------------------------------------------------------ Verify
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_verify_matches_reference_golden(ctx, golden_dir, mode):
    """K4 against Verify values produced by the reference build (exact integer inlier counts)."""
    g = np.load(os.path.join(golden_dir, "s4pcs_case1.npz"))
    # centred P exactly as MatchBase::init does it: sequential float sum, then subtract
    P = g["P_xyz"].astype(np.float32)
    cen = g["cP"]
    Pc = (P - cen).astype(np.float32)
    ctx.verify_set_clouds(Pc, g["Qs"])
    cnt = ctx.verify_batch(g["verify_T"], 0.003, mode)
    nq = len(g["Qs"])
    assert np.array_equal(cnt.astype(np.float32) / np.float32(nq), g["verify_lcp"])
    assert cnt.max() > 0


def test_verify_brute_grid_oracle_agree_large(ctx, orc, synth):
    """20k-point scene, 100 samples, 512 transforms: brute == grid == oracle (kd-tree), exact counts."""
    rng = np.random.default_rng(3)
    sc = synth.make_scene(20000, seed=7)
    mx, _ = synth.ellipsoid_model(5000)
    P = (sc.xyz - sc.xyz.mean(axis=0)).astype(np.float32)
    Qraw = mx[rng.choice(len(mx), 100, replace=False)]
    cq = Qraw.mean(axis=0)
    Qs = (Qraw - cq).astype(np.float32)
    Tg = sc.gt_pose.copy()
    Tg[:3, 3] = Tg[:3, :3] @ cq + Tg[:3, 3] - sc.xyz.mean(axis=0)   # centred-frame transform
    T = synth.replay_poses(Tg, 512, seed=5, max_rot_deg=20, max_trans=0.01)
    ctx.verify_set_clouds(P, Qs)
    a = ctx.verify_batch(T, 0.003, 0)
    b = ctx.verify_batch(T, 0.003, 1)
    assert np.array_equal(a, b)
    assert np.array_equal(a, ctx.verify_batch(T, 0.003, 2))
    # poses right at the inlier boundary (noise of the order of delta) and far-off poses (empty cells)
    T2 = synth.replay_poses(Tg, 2048, seed=9, max_rot_deg=60, max_trans=0.05)
    for delta in (0.003, 0.0011):
        r0 = ctx.verify_batch(T2, delta, 0)
        assert np.array_equal(r0, ctx.verify_batch(T2, delta, 1)) and np.array_equal(r0, ctx.verify_batch(T2, delta, 2))
    o = orc.verify_batch(P, Qs, T[:96], 0.003, use_tree=True)
    assert np.array_equal(a[:96], o)
    assert a.max() > 5
    # order independence of the scene: a size-independent property of Verify
    perm = rng.permutation(len(P))
    ctx.verify_set_clouds(P[perm], Qs)
    assert np.array_equal(ctx.verify_batch(T, 0.003, 0), a)
    assert np.array_equal(ctx.verify_batch(T, 0.003, 2), a)


# ------------------------------------------------------------------------------------------------ generator
@pytest.mark.parametrize("case", ["case1", "case2", "depth7"])
@pytest.mark.parametrize("mode", [0, 1, 2])
def test_generator_matches_reference_golden(ctx, api, orc, golden_dir, case, mode):
    g = np.load(os.path.join(golden_dir, f"s4pcs_{case}.npz"))
    sample_size, succ, n_calls = (int(v) for v in g["opts"])
    assert n_calls == 1
    overlap, delta, disp = (float(v) for v in g["opts_f"])
    ctx.set_scene(g["P_xyz"], g["P_nrm"], g["P_conf"], 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, g["Q_xyz"], g["Q_nrm"])
    ctx.set_ppf_keys(g["keys"])
    o = ctx.default_s4pcs_opts(sample_size=sample_size, overlap=overlap, delta=delta, dispersion=disp,
                               success_quadrilaterals=succ, max_time_seconds=0, n_trials=0, verify_mode=mode)
    pose, lcp, st = ctx.s4pcs_generate(o)
    # state
    assert st.n_sampled_q == len(g["Qs"])
    qs, qn = ctx.s4pcs_sampled_q(st.n_sampled_q)
    assert np.array_equal(qs, g["Qs"]) and np.array_equal(qn, g["Qs_nrm"])
    assert np.array_equal(np.array(st.centroid_p, np.float32), g["cP"])
    assert np.array_equal(np.array(st.centroid_q, np.float32), g["cQ"])
    assert np.float32(st.diameter) == g["diameter"]
    # per-base trace: ids, invariants bit-equal; list sizes equal to the reference's
    bases = ctx.s4pcs_bases()
    assert len(bases) == int(g["n_bases"])
    for i, b in enumerate(bases):
        assert np.array_equal(b["base"], g["base_ids"][i])
        assert np.array_equal(b["inv"], g["base_inv"][i])
        assert b["n_pairs1"] == len(g[f"pairs1_{i}"])
        assert b["n_pairs2"] == len(g[f"pairs2_{i}"])
        assert b["n_quads"] == len(g[f"quads_{i}"])
    # hypotheses: same multiset as the reference (lcp and rotation exact, translation 1e-6 m)
    assert len(lcp) == len(g["hyp_lcp"])
    p2, l2 = _canon(pose, lcp)
    assert np.array_equal(l2, g["hyp_lcp"])
    assert np.array_equal(p2[:, :3, :3], g["hyp_pose"][:, :3, :3])
    assert np.abs(p2[:, :3, 3] - g["hyp_pose"][:, :3, 3]).max() < 1e-6
    # emission order: identical to the oracle's canonical order, element by element
    oo = orc.OracleS4PCS(sample_size=sample_size, overlap=overlap, delta=delta, dispersion=disp, success_quadrilaterals=succ)
    oo.set_keys(g["keys"])
    oo.run(g["P_xyz"], g["P_nrm"], g["P_conf"], g["Q_xyz"], g["Q_nrm"], 1)
    op, ol = oo.hypos()
    assert np.array_equal(ol, lcp)
    assert np.array_equal(op[:, :3, :3], pose[:, :3, :3])
    assert np.abs(op[:, :3, 3] - pose[:, :3, 3]).max() < 1e-6


def test_generator_explicit_trials_matches_oracle(ctx, api, orc, synth):
    """n_trials as an explicit parameter (beyond the reference's 30-trial clamp), larger scene."""
    mx, mn = synth.ellipsoid_model_spacing(0.005)
    keys = synth.ppf_key_table()
    sc = synth.make_scene(2000, seed=13)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_ppf_keys(keys)
    o = ctx.default_s4pcs_opts(sample_size=80, success_quadrilaterals=1000, max_time_seconds=0, n_trials=48)
    pose, lcp, st = ctx.s4pcs_generate(o, cap=1 << 18)
    oo = orc.OracleS4PCS(sample_size=80, success_quadrilaterals=1000, n_trials=48)
    oo.set_keys(keys)
    n = oo.run(sc.xyz, sc.nrm, sc.conf, mx, mn, 1)
    op, ol = oo.hypos()
    assert st.n_trials_run == 48
    assert len(lcp) == n
    assert np.array_equal(ol, lcp)
    assert np.array_equal(op[:, :3, :3], pose[:, :3, :3])
    ob = oo.bases()
    gb = ctx.s4pcs_bases()
    assert len(ob) == len(gb)
    for a, b in zip(ob, gb):
        assert np.array_equal(a["base"], b["base"]) and np.array_equal(a["inv"], b["inv"])
        assert len(a["pairs1"]) == b["n_pairs1"] and len(a["pairs2"]) == b["n_pairs2"] and len(a["quads"]) == b["n_quads"]


# ------------------------------------------------------------------------------------------------ computeLCP
def _scoring_case(synth, n_scene, n_model, H, seed=7):
    sc = synth.make_scene(n_scene, seed=seed)
    mx, mn = synth.ellipsoid_model(n_model)
    poses = synth.replay_poses(sc.gt_pose, H, seed=11)
    poses[0] = sc.gt_pose.astype(np.float32)
    return sc, mx, mn, poses


@pytest.mark.parametrize("nn_mode", [0, 1, 2])
def test_lcp_scores_bit_equal_to_oracle(ctx, api, orc, synth, nn_mode):
    sc, mx, mn, poses = _scoring_case(synth, 3000, 2500, 48)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    ctx.hypos_upload(poses)
    best_pose, best_score, best_idx = ctx.lcp_select_best(0.001, 10.0, nn_mode)
    _, scores, _ = ctx.hypos_download()
    ref = orc.compute_lcp_batch(sc.xyz, sc.nrm, mx, mn, poses, 0.001, 10.0, use_tree=True)
    assert scores.max() > 10.0, "the ground-truth pose must score"
    assert np.array_equal(scores, ref), (scores[:6], ref[:6])
    # selectBest: first strict maximum
    exp = int(np.flatnonzero(ref == ref.max())[0])
    assert best_idx == exp and best_score == ref[exp]
    assert np.array_equal(best_pose, poses[exp])


@pytest.mark.parametrize("nn_mode", [0, 1, 2])
def test_lcp_ragged_and_empty_edge_cases(ctx, api, orc, synth, nn_mode):
    """Sizes that are not multiples of any tile (1 scene point .. 2049 model points), and a pose far away."""
    sc, mx, mn, poses = _scoring_case(synth, 777, 2049, 5)
    far = poses[1].copy()
    far[:3, 3] += 10.0
    poses[1] = far
    for ns in (1, 63, 777):
        ctx.set_scene(sc.xyz[:ns], sc.nrm[:ns], None, 0.0)
        ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
        ctx.hypos_upload(poses)
        ctx.lcp_select_best(0.002, 15.0, nn_mode)
        _, scores, _ = ctx.hypos_download()
        ref = orc.compute_lcp_batch(sc.xyz[:ns], sc.nrm[:ns], mx, mn, poses, 0.002, 15.0, use_tree=False)
        assert np.array_equal(scores, ref)
        assert scores[1] == 0.0


# ------------------------------------------------------------------------------------------------ ICP
@pytest.mark.parametrize("nn_mode", [0, 1, 2, 3])
def test_icp_matches_oracle_and_recovers_pose(ctx, api, orc, synth, nn_mode):
    sc, mx, mn, poses = _scoring_case(synth, 2500, 1500, 24)
    # small perturbations so that ICP converges to the truth
    poses = synth.replay_poses(sc.gt_pose, 24, seed=3, max_rot_deg=6.0, max_trans=0.004)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.hypos_upload(poses)
    it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=nn_mode, want_stats=True)
    out, _, _ = ctx.hypos_download()
    ref, rit, rcv = orc.icp_refine_batch(sc.xyz, sc.nrm, mx, mn, poses, 10, 45.0, 0.01, use_tree=True)
    assert np.array_equal(it, rit) and np.array_equal(cv, rcv)
    assert np.abs(out - ref).max() < 2e-5          # float tolerance of the double-precision solve
    # refined poses are within 1 mm / 1 deg of the ground truth (modulo the ellipsoid's 180 deg symmetries
    # no symmetry flip can happen from a 6 deg start)
    gt = sc.gt_pose
    ok = 0
    for T in out:
        if np.linalg.norm(T[:3, 3] - gt[:3, 3]) < 1e-3 and _rot_err_deg(T[:3, :3].astype(np.float64), gt[:3, :3]) < 1.0:
            ok += 1
    assert ok >= 20


@pytest.mark.parametrize("nn_mode", [0, 1, 2, 3])
def test_icp_too_few_correspondences_returns_input(ctx, api, orc, synth, nn_mode):
    sc, mx, mn, poses = _scoring_case(synth, 500, 400, 3)
    poses[:, :3, 3] += 5.0  # nothing within 1 cm -> not converged -> identity (Utils.cpp:218-225)
    ctx.set_scene(sc.xyz, sc.nrm, None, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.hypos_upload(poses)
    it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=nn_mode, want_stats=True)
    out, _, _ = ctx.hypos_download()
    assert not cv.any() and not it.any()
    assert np.abs(out - poses).max() < 1e-6


def test_icp_and_lcp_grid_equal_brute_bitwise_large(ctx, api, synth):
    """C2-sized clouds, wide perturbations (far points exercise the ring expansion): the grid path must return
    the very same bits as the brute-force path (same correspondences -> same sums -> same poses / scores)."""
    sc, mx, mn, _ = _scoring_case(synth, 20000, 5000, 4)
    poses = synth.replay_poses(sc.gt_pose, 96, seed=21, max_rot_deg=40.0, max_trans=0.02)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    res = []
    for mode in (0, 1, 2, 3):
        ctx.hypos_upload(poses)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
        p, _, _ = ctx.hypos_download()
        ctx.lcp_select_best(0.001, 10.0, min(mode, 2))   # computeLCP nn_mode 3 is the reduced-sum mode (own test)
        _, sc_, _ = ctx.hypos_download()
        res.append((it.copy(), cv.copy(), p.copy(), sc_.copy()))
    # mode 1 walks the scene in the caller's order like mode 0: bit-identical.  Mode 2 walks it in Morton order, so
    # its f64 normal-equation sums associate differently: same iteration counts, poses equal to float rounding, and
    # the computeLCP scores (summed in the caller's order in every mode) equal for equal poses.
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    assert np.array_equal(res[0][0], res[2][0]) and np.array_equal(res[0][1], res[2][1])
    assert np.abs(res[0][2] - res[2][2]).max() < 2e-6
    same = np.all(res[0][2] == res[2][2], axis=(1, 2))
    assert same.mean() > 0.5 and np.array_equal(res[0][3][same], res[2][3][same])
    assert res[0][0].max() >= 4, "some hypotheses must need several iterations"
    # mode 3 (search and accumulation fused) visits the points in the same order as mode 2: the very same bits
    assert np.array_equal(res[2][0], res[3][0]) and np.array_equal(res[2][2], res[3][2]) and np.array_equal(res[2][3], res[3][3])
    # computeLCP alone on identical poses (the refined set of mode 0): all three NN modes return the same bits
    lcp = []
    for mode in (0, 1, 2):
        ctx.hypos_upload(res[0][2])
        ctx.lcp_select_best(0.001, 10.0, mode)
        lcp.append(ctx.hypos_download()[1].copy())
    assert np.array_equal(lcp[0], lcp[1]) and np.array_equal(lcp[0], lcp[2])
    assert (lcp[0] > 0).sum() > 10


# ------------------------------------------------------------------------------------------------ resident set
def test_topk_and_cluster(ctx, api, orc, synth):
    rng = np.random.default_rng(0)
    sc, mx, mn, poses = _scoring_case(synth, 300, 300, 400)
    scores = (rng.integers(0, 20, size=len(poses)) / 20.0).astype(np.float32)  # many ties
    ctx.hypos_upload(poses, scores)
    ctx.hypos_keep_topk(150)
    p, s, ids = ctx.hypos_download()
    order = np.lexsort((np.arange(len(scores)), -scores))[:150]
    assert np.array_equal(s, scores[order]) and np.array_equal(p, poses[order])
    assert np.array_equal(ids, np.arange(150))
    # clustering: device-resident call == pure host function == oracle
    ctx.hypos_upload(poses, scores)
    ctx.cluster_poses(30.0, 0.015, [180, 180, 180], True)
    pc, scs, idc = ctx.hypos_download()
    keep = orc.cluster_poses(poses, scores, np.arange(len(poses)), 30.0, 0.015, [180, 180, 180])
    keep2 = api.cluster_poses_host(poses, scores, np.arange(len(poses)), 30.0, 0.015, [180, 180, 180])
    assert np.array_equal(keep, keep2)
    assert np.array_equal(pc, poses[keep]) and np.array_equal(scs, scores[keep])
    assert np.array_equal(idc, np.arange(len(keep)))
    rows, n = ctx.topk_pack(8, id_offset=1000)
    pp, ss, ii = api.rows_to_hypos(rows)
    assert n == min(8, len(keep)) and (ii[:n] >= 1000).all()


# ------------------------------------------------------------------------------------------------ hand
def _hand_case(synth, api, n_scene=4000):
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    xyz, nrm = synth.make_hand_scene(hand, true, n_scene, seed=5)
    return hand, true, xyz, nrm


def _cfg():
    return {"hand_match": {"finger1_min_match": 5, "finger2_min_match": 5, "finger1_dist_thres": 0.005, "finger2_dist_thres": 0.005,
                           "finger1_normal_angle": 60, "finger2_normal_angle": 60, "check_normal": True, "max_outter_pts": 300,
                           "outter_pt_dist": 0.002, "outter_pt_dist_weight": 1, "planar_dist_thres": 0.001,
                           "pso": {"n_pop": 15, "n_gen": 3, "check_freq": 10, "pso_par_c_cog": 0.1, "pso_par_c_soc": 0.9,
                                   "pso_par_initial_w": 0.0}}}


def _oracle_args(orc, a, xyz, nrm, swivel, keep):
    """Copies a hop FingerArgs into the oracle's struct."""
    o = orc.FingerArgs()
    for k in range(3):
        o.fp_min[k], o.fp_max[k], o.fo_min[k], o.fo_max[k] = a.fp_min[k], a.fp_max[k], a.fo_min[k], a.fo_max[k]
    o.fp_stride_z, o.fp_num_division, o.fp_hist_min_y = a.fp_stride_z, a.fp_num_division, a.fp_hist_min_y
    for k in range(16):
        o.model2handbase[k], o.finger_out2parent[k] = a.model2handbase[k], a.finger_out2parent[k]
    for k in range(4):
        o.pair_tip1[k], o.pair_tip2[k] = a.pair_tip1[k], a.pair_tip2[k]
    for f in ("is_palm_side", "is_right_side", "gripper_min_dist", "dist_thres", "cos_normal_thres", "check_normal",
              "max_outter_pts", "outter_pt_dist", "outter_pt_dist_weight", "n_model"):
        setattr(o, f, getattr(a, f))
    o.model_xyz, o.model_nrm = a.model_xyz, a.model_nrm
    S, Ln, W = orc.soa(xyz), orc.soa(nrm), orc.soa(swivel)
    keep.extend([S, Ln, W])
    o.scene_xyz, o.n_scene = orc.F(S), S.shape[1]
    o.scene_nrm_lookup, o.n_lookup = orc.F(Ln), Ln.shape[1]
    o.swivel_xyz, o.n_swivel = orc.F(W), W.shape[1]
    return o


@pytest.mark.parametrize("finger", ["finger_1_1", "finger_2_2"])
def test_pso_objective_equals_oracle(ctx, api, orc, synth, finger):
    import ctypes as C
    hand, true, xyz, nrm = _hand_case(synth, api)
    swivel = xyz[xyz[:, 0] < -0.1]  # scene_remove_swivel: x in (-0.25,-0.1), Hand.cpp:316-320
    h = api.HandT42(_cfg(), hand, ctx=ctx)
    h.gripper_min_dist = 0.0144
    if finger.endswith("_2"):   # the proximal link is already placed when the distal one is searched
        par = finger[:-1] + "1"
        a = np.float32(true[par])
        T = np.eye(4, dtype=np.float32)
        T[1, 1], T[1, 2], T[2, 1], T[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        h._tf_self[par] = T
    h.setCurScene(xyz, nrm, swivel)
    args = h.finger_args(finger, 0.005)
    ctx.hand_set_finger(args)
    angles = np.radians(np.linspace(0.0, 120.0 if finger.endswith("_1") else 90.0, 97))
    got = ctx.hand_pso_eval_batch(angles)
    keep = []
    oa = _oracle_args(orc, args, xyz, nrm, swivel, keep)
    ref = np.zeros(len(angles))
    orc.lib().orc_pso_objective_batch(C.byref(oa), orc.D(np.ascontiguousarray(angles)), len(angles), orc.D(ref))
    assert np.array_equal(got, ref), np.abs(got - ref).max()
    # hop_hand_set_sum_mode(1): the outer-side penalty reduced in a tree instead of in scene order -- same objective to
    # 1e-5 relative (the penalty enters through avg = sum / count and an exp), same arg-min
    ctx.hand_set_sum_mode(1)
    try:
        fast = ctx.hand_pso_eval_batch(angles)
    finally:
        ctx.hand_set_sum_mode(0)
    assert np.all(np.abs(fast - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0)), np.abs(fast - ref).max()
    assert np.argmin(fast) == np.argmin(ref)
    # the objective has a real minimum near the true angle
    best = angles[np.argmin(ref)]
    assert abs(best - true[finger]) < math.radians(8.0)
    assert ref.min() < -5


def test_pso_search_equals_oracle(ctx, api, orc, synth):
    import ctypes as C
    hand, true, xyz, nrm = _hand_case(synth, api)
    swivel = xyz[xyz[:, 0] < -0.1]  # scene_remove_swivel: x in (-0.25,-0.1), Hand.cpp:316-320
    h = api.HandT42(_cfg(), hand, ctx=ctx)
    h.gripper_min_dist = 0.0144
    h.setCurScene(xyz, nrm, swivel)
    ok = h.matchOneComponentPSO("finger_1_1", 0, 120, False, 0.005, 60, 5)
    assert ok
    args = h.finger_args("finger_1_1", 0.005)
    # note: _tf_self[finger_1_1] is now set, but model2handbase of the link itself excludes its own angle?  No:
    # getTFHandBase includes _tf_self of the link, exactly as the reference does on a second call.
    h._tf_self["finger_1_1"] = np.eye(4, dtype=np.float32)
    args = h.finger_args("finger_1_1", 0.005)
    keep = []
    oa = _oracle_args(orc, args, xyz, nrm, swivel, keep)
    s = h.pso_settings(0, 120)
    os_ = orc.PsoSettings(s.n_pop, s.n_gen, s.check_freq, s.c_cog, s.c_soc, s.initial_w, s.w_min, s.w_max, s.err_tol, s.lower_rad,
                          s.upper_rad, s.seed)
    ang = C.c_double(0)
    val = C.c_double(0)
    orc.lib().orc_pso_search(C.byref(oa), C.byref(os_), C.byref(ang), C.byref(val))
    assert h.last_angle == np.float32(ang.value)
    assert h.last_objval == val.value
    assert abs(ang.value - true["finger_1_1"]) < math.radians(10.0)


# ------------------------------------------------------------------------------------------------ pipeline
def test_full_chain_pose_within_1mm_1deg_of_oracle(ctx, api, orc, synth):
    """gen -> cluster(30,15mm) -> ICP(<=100) -> cluster(5,3mm) -> selectBest, as main_realdata_auto.cpp:187-204
    (minus the two rejectBy* steps, which are 'next' rows): GPU chain vs oracle chain on the same inputs -- with ICP nn_mode 0, the
    ONE-GAUSS-NEWTON-STEP variant of rounds 1-2 against the oracle's variant of the same (NOT the reference's minimiser: a faster option the
    library keeps).  The chain with the minimiser the reference runs and the mirrors ship (nn_mode 7) is asserted bit for bit in
    tests/test_gpu_zy_icp_canon.py (this scene: test_ellipse_chain_returns_the_oracle_chains_pose_and_the_truth)."""
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    sc = synth.make_scene(1200, seed=7)
    sym = [180, 180, 180]
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(keys)
    o = ctx.default_s4pcs_opts(max_time_seconds=0)
    pose, lcp, st = ctx.s4pcs_generate(o)
    ctx.cluster_poses(30.0, 0.015, sym, True)
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0)
    # oracle chain
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(sc.xyz, sc.nrm, sc.conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert np.array_equal(ol, lcp)
    keep = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, sym)
    p1, l1 = op[keep][:100], ol[keep][:100]
    p2, _, _ = orc.icp_refine_batch(sc.xyz, sc.nrm, mx5, mn5, p1, 10, 45.0, 0.01)
    keep2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, sym)
    p3 = p2[keep2]
    s3 = orc.compute_lcp_batch(sc.xyz, sc.nrm, mx1, mn1, p3, 0.001, 10.0)
    ob = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    assert np.linalg.norm(best[:3, 3] - ob[:3, 3]) < 1e-3
    assert _rot_err_deg(best[:3, :3].astype(np.float64), ob[:3, :3].astype(np.float64)) < 1.0
    assert abs(score - s3.max()) <= 1e-4 * s3.max()
    # and the estimate is a correct pose: the model under `best` lies on the model under the truth
    a = mx1 @ best[:3, :3].T + best[:3, 3]
    b = mx1 @ sc.gt_pose[:3, :3].T + sc.gt_pose[:3, 3]
    from scipy.spatial import cKDTree
    adi = cKDTree(b).query(a)[0].mean()
    assert adi < 0.005  # the authors' evaluator threshold (scripts/eval_all.py:77)


def test_library_fails_loudly_without_fallback(api):
    with pytest.raises(api.HopError):
        api.Context(99)  # no such device: an error code, not a CPU path


# ------------------------------------------------------------------------------------------------ C++ host
def _write_cloud(path, xyz, nrm=None, conf=None):
    """frame-dir cloud format of host/app/main_realdata_auto.cpp: int32 n, int32 has_conf, xyz planes, normal planes, [conf]"""
    xyz = np.asarray(xyz, np.float32)
    nrm = np.zeros_like(xyz) if nrm is None else np.asarray(nrm, np.float32)
    with open(path, "wb") as f:
        np.array([len(xyz), 0 if conf is None else 1], np.int32).tofile(f)
        np.ascontiguousarray(xyz.T).tofile(f)
        np.ascontiguousarray(nrm.T).tofile(f)
        if conf is not None:
            np.asarray(conf, np.float32).tofile(f)


def test_cpp_host_app_equals_python_mirror(api, hop, synth, tmp_path):
    """The C++ host above the C-ABI (host/app/main_realdata_auto.cpp: the reference driver's call order,
    main_realdata_auto.cpp:99-205) and the Python mirror run the same frame: same finger angles, same best pose."""
    import subprocess
    from conftest import config_for_run
    cfg_path = config_for_run(os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml"), tmp_path)
    from hop_amd import config as hop_config
    cfg = hop_config.load_config(cfg_path)
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "main_realdata_auto")
    assert os.path.exists(exe), "build() must have produced the host application"
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    sc = synth.make_scene(1200, seed=7)
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    hxyz, hnrm = synth.make_hand_scene(hand, true, 4000, seed=5)
    swivel = hxyz[hxyz[:, 0] < -0.1]
    frame = tmp_path / "frame"
    out = tmp_path / "out"
    frame.mkdir()
    out.mkdir()
    _write_cloud(frame / "model.bin", mx5, mn5)
    _write_cloud(frame / "model001.bin", mx1, mn1)
    _write_cloud(frame / "object_segment.bin", sc.xyz, sc.nrm, sc.conf)
    with open(frame / "ppf_keys.bin", "wb") as f:
        np.array([len(keys)], np.int32).tofile(f)
        np.ascontiguousarray(keys, np.int32).tofile(f)
    with open(frame / "hand.txt", "w") as f:
        for name in hand.clouds:
            if name == "base_link":
                continue
            x, n = hand.clouds[name]
            _write_cloud(frame / f"{name}.bin", x, n)
            f.write(f"{name} {hand.parents[name]} {name}.bin " + " ".join(repr(float(v)) for v in hand.tf_in_parent[name].reshape(16)) + "\n")
    _write_cloud(frame / "hand_scene.bin", hxyz)
    _write_cloud(frame / "hand_region.bin", hxyz, hnrm)
    _write_cloud(frame / "hand_swivel.bin", swivel)
    (frame / "cam_side.txt").write_text("1\n")
    r = subprocess.run([exe, cfg_path, str(frame), str(out)], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout + r.stderr
    cpp_pose = np.loadtxt(out / "model2scene.txt").astype(np.float32)
    cpp_angles = {ln.split()[0]: float(ln.split()[1]) for ln in (out / "finger_angles.txt").read_text().splitlines()}

    est = api.PoseEstimator(cfg, (mx5, mn5), (mx1, mn1))
    h = api.HandT42(cfg, hand, ctx=est.ctx)
    ext = np.abs(mx1.min(axis=0) - mx1.max(axis=0))
    h.gripper_min_dist = 0.8 * float(ext.min())  # main_realdata_auto.cpp:41-45
    h.setCurScene(hxyz, hnrm, swivel)
    hm = cfg["hand_match"]
    py_angles = {}
    for first, second in (("finger_2_1", "finger_2_2"), ("finger_1_1", "finger_1_2")):
        if h.matchOneComponentPSO(first, 0, 120, False, hm["finger1_dist_thres"], hm["finger1_normal_angle"], hm["finger1_min_match"]):
            py_angles[first] = h.last_angle
            if h.matchOneComponentPSO(second, 0, 90, True, hm["finger2_dist_thres"], hm["finger2_normal_angle"], hm["finger2_min_match"]):
                py_angles[second] = h.last_angle
    est.setCurScene(sc.xyz, sc.nrm, sc.conf)
    assert est.runSuper4pcs(keys)
    est.clusterPoses(30, 0.015, True)
    est.refineByICP()
    est.clusterPoses(5, 0.003, False)
    best = est.selectBest()
    assert set(cpp_angles) == set(py_angles) and len(py_angles) >= 2
    for k, v in py_angles.items():
        assert abs(cpp_angles[k] - v) < 1e-6, (k, cpp_angles[k], v)
    assert np.abs(cpp_pose - best._pose).max() < 1e-6
    a = mx1 @ cpp_pose[:3, :3].T + cpp_pose[:3, 3]
    b = mx1 @ sc.gt_pose[:3, :3].T + sc.gt_pose[:3, 3]
    from scipy.spatial import cKDTree
    assert cKDTree(b).query(a)[0].mean() < 0.005


def test_c1_depth7_full_chain_within_1mm_1deg_of_oracle(ctx, api, orc, synth, golden_dir):
    """BASELINE.json configs[0] ("C1-synthetic-model"): the hand-region cloud of the reference's
    example/depth7.png (fixture made by tools/make_depth7_fixture.py: back-projection, hand-base crop, 3 mm voxels)
    with the as-shipped option values and the synthetic ellipse.  The north star's criterion: the pose returned by the
    GPU chain within 1 mm / 1 degree of the CPU restatement's on the same frame -- here with ICP nn_mode 3 (ONE GAUSS-NEWTON STEP per
    iteration, the variant of rounds 1-2, against the oracle's variant of the same; NOT the reference's minimiser).  With the minimiser the
    reference runs (nn_mode 7, what the mirrors ship) the same frame is asserted bit for bit in tests/test_gpu_zy_icp_canon.py
    (test_c1_depth7_chain_returns_the_oracle_chains_pose), which also says what can and cannot be expected of C1 against Eigen's own run."""
    g = np.load(os.path.join(golden_dir, "depth7_hand_region.npz"))
    xyz, nrm = g["xyz"], g["nrm"]
    conf = np.ones(len(xyz), np.float32)
    assert 1500 < len(xyz) < 2100 and int(g["counts"][0]) == 68600  # valid depth pixels of the example frame
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    sym = [180, 180, 180]
    ctx.set_scene(xyz, nrm, conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(keys)
    o = ctx.default_s4pcs_opts(max_time_seconds=0)   # sample 100, overlap 0.2, delta 3 mm, 10 successful bases
    pose, lcp, st = ctx.s4pcs_generate(o)
    ctx.cluster_poses(30.0, 0.015, sym, True)
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=3)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, 2)
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(xyz, nrm, conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert len(ol) > 1000 and np.array_equal(ol, lcp)
    keep = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, sym)
    p1, l1 = op[keep][:100], ol[keep][:100]
    p2, _, _ = orc.icp_refine_batch(xyz, nrm, mx5, mn5, p1, 10, 45.0, 0.01)
    keep2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, sym)
    p3 = p2[keep2]
    s3 = orc.compute_lcp_batch(xyz, nrm, mx1, mn1, p3, 0.001, 10.0)
    ob = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    assert np.linalg.norm(best[:3, 3] - ob[:3, 3]) < 1e-3
    assert _rot_err_deg(best[:3, :3].astype(np.float64), ob[:3, :3].astype(np.float64)) < 1.0
    assert abs(score - s3.max()) <= 1e-4 * s3.max()


def test_c1_depth7_from_the_raw_depth_image(ctx, api, orc, synth, golden_dir):
    """The same C1 criterion, starting from the reference's raw example/depth7.png (tests/golden/depth7_raw.npz) instead of a
    prepared cloud: depth image -> organised cloud -> integral-image normals -> 1 mm voxel grid -> hand-base crop -> 3 mm
    voxel grid on the GPU (main_realdata_auto.cpp:54-96, Hand.cpp:285), then the as-shipped chain on the GPU and in the
    oracle on that cloud: best pose within 1 mm / 1 degree.  The front end itself against the oracle's front end."""
    g = np.load(os.path.join(golden_dir, "depth7_raw.npz"))
    lo, hi = (-0.25, -0.2, -0.12), (-0.07, 0.2, 0.05)
    sx, sn, counts = ctx.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    rx, rn = orc.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    assert counts[0] == 68600 and sx.shape == rx.shape and np.abs(sx - rx).max() < 1e-6
    m = np.isfinite(sn).all(axis=1)
    assert np.array_equal(m, np.isfinite(rn).all(axis=1)) and np.abs(sn[m] - rn[m]).max() < 5e-5
    xyz, nrm = ctx.voxel_downsample_normals(sx[m], sn[m], 0.003)     # (points without a defined normal dropped: depth edges)
    assert 1200 < len(xyz) < 2100
    conf = np.ones(len(xyz), np.float32)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    sym = [180, 180, 180]
    ctx.set_scene(xyz, nrm, conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(keys)
    pose, lcp, st = ctx.s4pcs_generate(ctx.default_s4pcs_opts(max_time_seconds=0))
    ctx.cluster_poses(30.0, 0.015, sym, True)
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=3)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, 2)
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(xyz, nrm, conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert len(ol) > 500 and np.array_equal(ol, lcp)
    keep = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, sym)
    p1, l1 = op[keep][:100], ol[keep][:100]
    p2, _, _ = orc.icp_refine_batch(xyz, nrm, mx5, mn5, p1, 10, 45.0, 0.01)
    keep2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, sym)
    p3 = p2[keep2]
    s3 = orc.compute_lcp_batch(xyz, nrm, mx1, mn1, p3, 0.001, 10.0)
    ob = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    assert np.linalg.norm(best[:3, 3] - ob[:3, 3]) < 1e-3
    assert _rot_err_deg(best[:3, :3].astype(np.float64), ob[:3, :3].astype(np.float64)) < 1.0


# ------------------------------------------------------------------------------------------------ next row N3a
def test_remove_surrounding_points_equals_oracle(api, orc, synth, hop):
    """HandT42::removeSurroundingPointsAndAssignProbability (Hand.cpp:779-888): same survivors as
    the CPU restatement, same coordinates bit for bit, confidences within one float ulp (expf vs exp-in-double)."""
    from hop_amd import config as hop_config
    cfg = hop_config.load_config()
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    hxyz, hnrm = synth.make_hand_scene(hand, true, 6000, seed=5)          # hand-base frame: hand surface + object blob
    rng = np.random.default_rng(3)
    extra = rng.uniform([-0.2, -0.1, -0.1], [0.0, 0.1, 0.05], size=(3000, 3)).astype(np.float32)   # free space around the hand
    hb_xyz = np.concatenate([hxyz, extra])
    hb_nrm = np.concatenate([hnrm, np.tile(np.float32([0, 0, 1]), (len(extra), 1))])
    handbase_in_cam = synth.se3(synth.rot_from_axis_angle(np.array([0.3, -0.5, 0.8]), 1.1), [0.05, -0.02, 0.45]).astype(np.float32)
    cam_xyz = (hb_xyz.astype(np.float64) @ handbase_in_cam[:3, :3].T.astype(np.float64) + handbase_in_cam[:3, 3]).astype(np.float32)
    cam_nrm = (hb_nrm.astype(np.float64) @ handbase_in_cam[:3, :3].T.astype(np.float64)).astype(np.float32)
    h = api.HandT42(cfg, hand)
    for name, ang in true.items():                                        # the finger state found by the hand search
        a = np.float32(ang)
        T = np.eye(4, dtype=np.float32)
        T[1, 1], T[1, 2], T[2, 1], T[2, 2] = np.cos(a), -np.sin(a), np.sin(a), np.cos(a)
        h._tf_self[name] = T
    near2 = float(np.float32(0.01) * np.float32(0.01))
    x, n, c, idx = h.removeSurroundingPointsAndAssignProbability(cam_xyz, cam_nrm, handbase_in_cam, near2)
    clouds = h.makeHandCloud()
    links = [(clouds[k], h.local_dist_thres(k, near2)) for k in sorted(clouds)]
    ox, on, oc, oidx = orc.hand_remove_surrounding(cam_xyz, cam_nrm, handbase_in_cam, links, h.getTFHandBase("finger_1_2"),
                                                   h.getTFHandBase("finger_2_2"), float(h._finger_properties["finger_1_2"]["min"][2]))
    assert np.array_equal(idx, oidx)
    assert 0.05 * len(cam_xyz) < len(idx) < 0.95 * len(cam_xyz)           # both outcomes are exercised
    assert np.array_equal(x, ox) and np.array_equal(n, on)
    ulp = np.spacing(np.maximum(np.abs(oc), np.float32(1e-30)))
    assert np.all(np.abs(c - oc) <= ulp)
    assert (np.abs(c - oc) == 0).mean() > 0.99
    assert c.min() > 0.5 and c.max() <= 1 and (c < 0.99).sum() > 10
    # the hand's own surface is gone, far free-space points stay with high confidence
    kept_hand = np.isin(np.arange(len(hxyz)), idx).mean()
    assert kept_hand < 0.7
    # empty input
    x0, n0, c0, i0 = h.removeSurroundingPointsAndAssignProbability(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), handbase_in_cam, near2)
    assert len(c0) == 0


def test_c5_scene_size_cell_lists_equal_brute_force(ctx, api, synth):
    """BASELINE.json configs[4] sizes per point cloud (50 000-point scene, 5000-point model), a handful of hypotheses:
    the cell-list paths (ICP nn_mode 3, computeLCP nn_mode 2, Verify mode 2) against the brute-force kernels."""
    sc = synth.make_scene(50000, seed=13)
    mx, mn = synth.ellipsoid_model(5000)
    poses = synth.replay_poses(sc.gt_pose, 48, seed=13, max_rot_deg=30.0, max_trans=0.015)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    out = {}
    for mode in (0, 3):
        ctx.hypos_upload(poses)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
        p, _, _ = ctx.hypos_download()
        out[mode] = (it.copy(), cv.copy(), p.copy())
    assert np.array_equal(out[0][0], out[3][0]) and np.array_equal(out[0][1], out[3][1])
    assert np.abs(out[0][2] - out[3][2]).max() < 2e-6
    scores = {}
    for mode in (0, 2):
        ctx.hypos_upload(out[0][2])
        ctx.lcp_select_best(0.001, 10.0, mode)
        scores[mode] = ctx.hypos_download()[1].copy()
    assert np.array_equal(scores[0], scores[2]) and scores[0].max() > 100
    rng = np.random.default_rng(5)
    P = (sc.xyz - sc.xyz.mean(axis=0)).astype(np.float32)
    Qs = mx[rng.choice(len(mx), 100, replace=False)]
    cq = Qs.mean(axis=0)
    Tg = sc.gt_pose.copy()
    Tg[:3, 3] = Tg[:3, :3] @ cq + Tg[:3, 3] - sc.xyz.mean(axis=0)
    T = synth.replay_poses(Tg, 256, seed=7, max_rot_deg=25, max_trans=0.01)
    ctx.verify_set_clouds(P, (Qs - cq).astype(np.float32))
    assert np.array_equal(ctx.verify_batch(T, 0.003, 0), ctx.verify_batch(T, 0.003, 2))


def test_recall_tool_gpu_matches_oracle_on_synthetic_frames():
    """tests/recall_eval.py (BASELINE metric, second half: ADD(-S) recall vs the CPU reference path) on a few frames of
    the synthetic substitute: every GPU pose within 1 mm / 1 degree of the oracle's, same recall."""
    import json
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "recall_eval.py"), "--frames", "6", "--scene", "1500"],
                       capture_output=True, text=True, timeout=6000 if os.environ.get("HOP_TEST_EMU") else 600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["frames_gpu_pose_within_1mm_1deg_of_cpu"] == 6
    assert out["icp_nn_mode"] == 7 and out["frames_gpu_pose_bit_equal_to_cpu"] == 6     # the shipped ICP mode returns the oracle's bits
    assert out["recall_adi_10mm_gpu"] == out["recall_adi_10mm_cpu"] and out["recall_adi_5mm_gpu"] == out["recall_adi_5mm_cpu"]
    assert out["recall_adi_10mm_gpu"] >= 0.8


def test_contexts_in_flight_are_independent(api, synth):
    """bench.py keeps several frames in flight, one hop_ctx and one host thread each: contexts share nothing, so frames
    run concurrently must return exactly what they return one at a time."""
    import threading
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    scenes = [synth.make_scene(3000, seed=40 + k) for k in range(3)]

    def frame(ctx, sc, out, slot):
        ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
        o = ctx.default_s4pcs_opts(success_quadrilaterals=64, n_trials=64, max_time_seconds=0)
        ctx.s4pcs_generate(o, download=False)
        ctx.hypos_keep_topk(512)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=3, want_stats=True)
        best, score, idx = ctx.lcp_select_best(0.001, 10.0, 2)
        pose, sc_, ids = ctx.hypos_download()
        out[slot] = (it.copy(), pose.copy(), sc_.copy(), best.copy(), score, idx)

    ctxs = []
    for _ in range(3):
        c = api.Context(0)
        c.set_model(api.HOP_MODEL_5MM, mx5, mn5)
        c.set_model(api.HOP_MODEL_1MM, mx1, mn1)
        c.set_ppf_keys(keys)
        ctxs.append(c)
    serial, conc = {}, {}
    for k in range(3):
        frame(ctxs[k], scenes[k], serial, k)
    for rep in range(2):
        ts = [threading.Thread(target=frame, args=(ctxs[(k + rep) % 3], scenes[k], conc, k)) for k in range(3)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for k in range(3):
            a, b = serial[k], conc[k]
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
            assert np.array_equal(a[3], b[3]) and a[4] == b[4] and a[5] == b[5]
    for c in ctxs:
        c.close()


def test_cell_lists_far_from_the_origin_equal_brute_force(ctx, api, synth):
    """The float error of the reference's distance expression grows with the coordinate magnitude; the list margins
    scale with it (hop_ctx::coord_mag).  A frame 6 m from the camera: cell-list paths against brute force."""
    sc = synth.make_scene(6000, seed=17)
    mx, mn = synth.ellipsoid_model(3000)
    shift = np.float32([1.5, -2.0, 5.5])
    xyz = (sc.xyz + shift).astype(np.float32)
    gt = sc.gt_pose.copy()
    gt[:3, 3] += shift
    poses = synth.replay_poses(gt, 64, seed=3, max_rot_deg=25.0, max_trans=0.01)
    ctx.set_scene(xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    out = {}
    for mode in (0, 3):
        ctx.hypos_upload(poses)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
        out[mode] = (it.copy(), cv.copy(), ctx.hypos_download()[0].copy())
    assert np.array_equal(out[0][0], out[3][0]) and np.array_equal(out[0][1], out[3][1])
    assert np.abs(out[0][2] - out[3][2]).max() < 2e-5   # float ulp at 6 m is 5e-7
    sc_ = {}
    for mode in (0, 2):
        ctx.hypos_upload(out[0][2])
        ctx.lcp_select_best(0.001, 10.0, mode)
        sc_[mode] = ctx.hypos_download()[1].copy()
    assert np.array_equal(sc_[0], sc_[2])


def test_model_ppf_keys_equal_oracle(ctx, orc, synth):
    """hop_model_ppf_keys (pair loop of the offline computePPF tool) against its CPU restatement:
    the same sorted key set; and the table it returns is usable by the generator as is."""
    mx, mn = synth.ellipsoid_model_spacing(0.005)
    k = ctx.model_ppf_keys(mx, mn * np.float32(2.5))          # unnormalised normals: normalised once, as the tool does
    o = orc.model_ppf_keys(mx, mn * np.float32(2.5))
    assert len(k) > 500 and np.array_equal(k, o)
    assert ctx.model_ppf_keys(mx[:1], mn[:1]).shape == (0, 4)
    with pytest.raises(Exception):
        ctx.model_ppf_keys(mx, mn, cap=10)                    # HOP_E_CAPACITY, not a silent truncation


def test_library_exchange_single_rank(api, ctx, synth):
    """hop_comm_* / hop_topk_allgather with a communicator of one rank (RCCL refuses two ranks on one device, so a single
    GPU box can exercise no more): the all-gather + merge returns the rank's own table."""
    if os.environ.get("HOP_TEST_EMU"):
        # (the CPU model answers RCCL through a stand-in library that must be on LD_LIBRARY_PATH when the PROCESS starts: this in-process
        # form cannot have it; the same calls run on the model in child processes -- tests/test_gpu_fullsize.py's exchange test, the
        # multi-rank tests of tests/test_emu_kernels_cpu.py)
        pytest.skip("in-process RCCL: not on the CPU model")
    uid = api.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = api.Comm(0, uid, 0, 1)
    sc, mx, mn, poses = _scoring_case(synth, 600, 500, 20)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    ctx.hypos_upload(poses)
    ctx.lcp_select_best(0.001, 10.0, 0)
    rows, n = ctx.topk_pack(8, id_offset=1000)
    merged, m = comm.topk_allgather(rows, 8)
    assert m == n == 8 and np.array_equal(merged, rows)
    for _ in range(3):      # repeated collectives on the communicator's stream
        assert np.array_equal(comm.topk_allgather(rows, 8)[0], rows)
    comm.close()


@pytest.mark.gpu
def test_ppf_matrix_symmetric_kernel_equals_direct_and_literal(api, synth, monkeypatch):
    """k_ppf_matrix_sym derives key(j -> i) from the registers of key(i -> j) (half the pair evaluations): the whole
    membership matrix must equal, bit for bit, the one of the direct threshold-bin kernel and the one of the literal
    acosf kernel -- on a scene whose size is not a multiple of 64 (ragged last word and row tile)."""
    import ctypes as C
    sc = synth.make_scene(3001, seed=11)
    mx, mn = synth.ellipsoid_model(2000)
    keys = synth.ppf_key_table()
    mats = {}
    for name, env in (("sym", {}), ("direct", {"HOP_PPF_NO_SYM": "1"}), ("literal", {"HOP_PPF_LITERAL": "1"})):
        for k in ("HOP_PPF_NO_SYM", "HOP_PPF_LITERAL"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = api.Context(0)
        try:
            c.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
            c.set_model(api.HOP_MODEL_5MM, mx, mn)
            c.set_ppf_keys(keys)
            o = c.default_s4pcs_opts(sample_size=60, success_quadrilaterals=4, max_time_seconds=0, n_trials=4, random_seed=5489)
            c.s4pcs_generate(o, cap=1 << 18)
            fn = c.L.hop_debug_ppf_matrix
            fn.restype = C.c_int
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.POINTER(C.c_int)]
            n, w = C.c_int(0), C.c_int(0)
            assert fn(c.h, None, 0, C.byref(n), C.byref(w)) == 0
            m = np.zeros((n.value, w.value), np.uint64)
            assert fn(c.h, m.ctypes.data_as(C.c_void_p), m.size, C.byref(n), C.byref(w)) == 0
            mats[name] = m
        finally:
            c.close()
    assert mats["sym"].shape[0] == 3001 and mats["sym"].any()
    assert np.array_equal(mats["sym"], mats["direct"])
    assert np.array_equal(mats["sym"], mats["literal"])
    # no bit on the diagonal, none beyond the last point
    n, w = mats["sym"].shape
    bits = np.unpackbits(mats["sym"].view(np.uint8).reshape(n, -1), axis=1, bitorder="little")
    assert not bits[:, n:].any() and not bits[np.arange(n), np.arange(n)].any()


@pytest.mark.gpu
def test_icp_duplicate_model_points_tie_to_the_lowest_index(ctx, api, synth):
    """Coincident model points with DIFFERENT normals: FLANN / the linear scan give a tie to the lowest original index, and
    the normal of that point enters the residual.  The packed lists of nn_mode 3 quantise both copies to the same entry
    coordinates and store them by Morton rank, so the tie must come out of the exact re-scan's original-index rule: the
    same bits as the plain cell lists (nn_mode 2), the brute-force iteration counts."""
    sc, mx, mn, _ = _scoring_case(synth, 6000, 1500, 9)
    rng = np.random.default_rng(5)
    tilt = mn + 0.2 * rng.normal(size=mn.shape).astype(np.float32)
    tilt /= np.linalg.norm(tilt, axis=1, keepdims=True)
    X = np.concatenate([mx, mx]).astype(np.float32)
    Nn = np.concatenate([mn, tilt]).astype(np.float32)
    perm = rng.permutation(len(X))                      # the copy with the lower index is sometimes the tilted one
    X, Nn = X[perm], Nn[perm]
    poses = synth.replay_poses(sc.gt_pose, 48, seed=3, max_rot_deg=15.0, max_trans=0.008)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, X, Nn)
    out = {}
    for mode in (0, 2, 3):
        ctx.hypos_upload(poses)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
        out[mode] = (it.copy(), cv.copy(), ctx.hypos_download()[0].copy())
    assert np.array_equal(out[0][0], out[2][0]) and np.abs(out[0][2] - out[2][2]).max() < 2e-6
    assert np.array_equal(out[2][0], out[3][0]) and np.array_equal(out[2][1], out[3][1]) and np.array_equal(out[2][2], out[3][2])
    # the tilted normals matter: with the clean normals alone the poses come out differently
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.hypos_upload(poses)
    ctx.icp_refine(10, 45.0, 0.01, nn_mode=3)
    assert np.abs(ctx.hypos_download()[0] - out[3][2]).max() > 1e-5
