"""GPU parity tests at the sizes BASELINE.json's configs name (C2: 20 000-point scene / 5 000-point model / 128-base
batches; C5: 50 000-point scene x 8 192 hypotheses per GPU), against the CPU oracle where it finishes in seconds and
through size-independent properties beyond that.  Run with `pytest -m gpu` on an MI355X."""
import math
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api(hop):
    from hop_amd import api as _api
    _api.lib()
    return _api


@pytest.fixture(scope="module")
def synth(hop):
    return hop.synth


@pytest.fixture()
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


def _rot_err_deg(Ra, Rb):
    c = (np.trace(Ra.T.astype(np.float64) @ Rb.astype(np.float64)) - 1) / 2
    return math.degrees(math.acos(max(-1.0, min(1.0, float(c)))))


def test_c2_generator_full_size_matches_oracle(ctx, api, orc, synth):
    """The generator exactly as bench.py runs it -- 20 000-point scene, 5 000-point model, sample_size 100, explicit
    trials shipped in 128-base batches (no early stop), the threshold PPF bins and the AVX-512 selection fast path --
    against the oracle: 256 base trials, every base, every list size, every hypothesis."""
    sc = synth.make_scene(20000, seed=7)
    mx, mn = synth.ellipsoid_model(5000)
    keys = synth.ppf_key_table()
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_ppf_keys(keys)
    T = 256
    o = ctx.default_s4pcs_opts(sample_size=100, success_quadrilaterals=T, max_time_seconds=0, n_trials=T, random_seed=5489)
    pose, lcp, st = ctx.s4pcs_generate(o, cap=1 << 21)
    keep = sc.conf >= 0.8
    oo = orc.OracleS4PCS(sample_size=100, success_quadrilaterals=T, n_trials=T, random_seed=5489)
    oo.set_keys(keys)
    n = oo.run(sc.xyz[keep], sc.nrm[keep], sc.conf[keep], mx, mn, 1)
    op, ol = oo.hypos()
    # (the reference's loop leaves at i / n_trials >= 0.99, congruentSetExplorationBase.hpp:177-186: 255 of 256 trials)
    assert st.n_trials_run >= 250 and len(lcp) == n and n > 10000
    assert np.array_equal(ol, lcp)
    assert np.array_equal(op[:, :3, :3], pose[:, :3, :3])
    assert np.abs(op[:, :3, 3] - pose[:, :3, 3]).max() < 1e-6
    ob, gb = oo.bases(), ctx.s4pcs_bases()
    assert len(ob) == len(gb) and len(gb) > 100
    for a, b in zip(ob, gb):
        assert np.array_equal(a["base"], b["base"]) and np.array_equal(a["inv"], b["inv"])
        assert len(a["pairs1"]) == b["n_pairs1"] and len(a["pairs2"]) == b["n_pairs2"] and len(a["quads"]) == b["n_quads"]


def test_c2_icp_and_lcp_full_size_match_oracle(ctx, api, orc, synth):
    """refineByICP and computeLCP at C2's cloud sizes on 64 hypotheses against the oracle (kd-tree NN): the default
    modes of the bench (ICP nn_mode 4 = composed increments, computeLCP nn_mode 3 = reduced sums) within their stated
    tolerances, the exact modes (3 / 2) bit-for-bit in iterations and scores."""
    sc = synth.make_scene(20000, seed=7)
    mx, mn = synth.ellipsoid_model(5000)
    keep = sc.conf >= 0.8
    S, Sn = sc.xyz[keep], sc.nrm[keep]
    poses = synth.replay_poses(sc.gt_pose, 64, seed=17, max_rot_deg=12.0, max_trans=0.006)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    ref, rit, rcv = orc.icp_refine_batch(S, Sn, mx, mn, poses, 10, 45.0, 0.01, use_tree=True)
    ctx.hypos_upload(poses)
    it3, cv3 = ctx.icp_refine(10, 45.0, 0.01, nn_mode=3, want_stats=True)
    p3, _, _ = ctx.hypos_download()
    assert np.array_equal(it3, rit) and np.array_equal(cv3, rcv)
    assert np.abs(p3 - ref).max() < 2e-5
    ctx.hypos_upload(poses)
    it4, cv4 = ctx.icp_refine(10, 45.0, 0.01, nn_mode=4, want_stats=True)
    p4, _, _ = ctx.hypos_download()
    assert (it4 == rit).mean() >= 0.95 and np.array_equal(cv4, rcv)
    for a, b in zip(p4, ref):   # north_star's tolerance: 1 mm / 1 degree
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) < 1e-3 and _rot_err_deg(a[:3, :3], b[:3, :3]) < 1.0
    assert np.median(np.abs(p4 - ref).reshape(len(ref), -1).max(axis=1)) < 2e-5
    # computeLCP on the refined poses
    sref = orc.compute_lcp_batch(S, Sn, mx, mn, p3, 0.001, 10.0, use_tree=True)
    ctx.hypos_upload(p3)
    _, s2, i2 = ctx.lcp_select_best(0.001, 10.0, 2)
    sc2 = ctx.hypos_download()[1].copy()
    assert np.array_equal(sc2, sref) and sref.max() > 1000
    ctx.hypos_upload(p3)
    _, s3, i3 = ctx.lcp_select_best(0.001, 10.0, 3)
    sc3 = ctx.hypos_download()[1].copy()
    big = sref > 1.0
    assert np.abs(sc3[big] - sref[big]).max() / sref.max() < 1e-4     # 1e-4 relative
    assert np.all(np.abs(sc3[big] - sref[big]) <= 1e-4 * sref[big])
    assert i3 == i2 == int(np.argmax(sref))


def test_icp_composed_increments_close_to_chained(ctx, api, synth):
    """nn_mode 4 applies the accumulated transform once (fused multiply-adds) instead of chaining the increments as PCL /
    the oracle do (nn_mode 3), and keeps the per-lane sums of the normal equations in float (<= ~30 terms per lane; lanes,
    waves and blocks are summed in double): a wide hypothesis set must keep its iteration counts (>= 99 %) and its poses
    (median <= 1e-5 -- micrometres --, every pose within 1 mm / 1 degree)."""
    sc = synth.make_scene(20000, seed=7)
    mx, mn = synth.ellipsoid_model(5000)
    poses = synth.replay_poses(sc.gt_pose, 512, seed=23, max_rot_deg=30.0, max_trans=0.015)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    out = {}
    for mode in (3, 4):
        ctx.hypos_upload(poses)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True)
        out[mode] = (it.copy(), cv.copy(), ctx.hypos_download()[0].copy())
    assert (out[3][0] == out[4][0]).mean() >= 0.99 and np.array_equal(out[3][1], out[4][1])
    d = np.abs(out[3][2] - out[4][2]).reshape(len(poses), -1).max(axis=1)
    assert np.median(d) <= 1e-5
    for a, b in zip(out[3][2], out[4][2]):
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) < 1e-3 and _rot_err_deg(a[:3, :3], b[:3, :3]) < 1.0


def test_c5_full_size_8192_hypotheses_50k_scene(ctx, api, synth):
    """BASELINE.json configs[4], one GPU's share: 8 192 replay hypotheses (seed 13) x 50 000-point scene x 5 000-point
    model through ICP and computeLCP in the bench's default modes.  A 256-hypothesis subsample is compared with the
    brute-force kernels (ICP nn_mode 0, computeLCP nn_mode 0); the whole set through properties: idempotence of the
    scoring, the reduced sums against the ordered sums (1e-4), score order (refined poses beat their starts), and the
    best pose within 1 mm / 1 degree of the ground truth."""
    sc = synth.make_scene(50000, seed=13)
    mx, mn = synth.ellipsoid_model(5000)
    H = 8192
    poses = synth.replay_poses(sc.gt_pose, H, seed=13, max_rot_deg=30.0, max_trans=0.015)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    ctx.hypos_upload(poses)
    it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=4, want_stats=True)
    refined = ctx.hypos_download()[0].copy()
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, 3)
    s3 = ctx.hypos_download()[1].copy()
    assert cv.mean() > 0.9 and it.max() <= 10
    # idempotence: scoring the same resident set again returns the same bits
    best_b, score_b, idx_b = ctx.lcp_select_best(0.001, 10.0, 3)
    assert idx_b == idx and score_b == score and np.array_equal(ctx.hypos_download()[1], s3)
    # reduced sums vs the reference's ordered sums, all 8192
    ctx.lcp_select_best(0.001, 10.0, 2)
    s2 = ctx.hypos_download()[1].copy()
    big = s2 > 1.0
    assert np.all(np.abs(s3[big] - s2[big]) <= 1e-4 * s2[big])
    # (8 192 starts converge to the same pose: hundreds of scores agree to 1e-5, and the arg-max of the reduced sums may be another member of
    # that tie than the arg-max of the ordered sums -- its ordered-sum score is within 1e-4 of the maximum)
    assert s2[idx] >= (1 - 1e-4) * s2.max()
    # the winner is the ground truth (modulo the ellipsoid's 180 degree symmetries)
    gt = sc.gt_pose.astype(np.float64)
    flips = [np.diag(v) for v in ([1, 1, 1], [1, -1, -1], [-1, 1, -1], [-1, -1, 1])]
    err = min(_rot_err_deg(best[:3, :3], gt[:3, :3] @ F) for F in flips)
    assert err < 1.0 and np.linalg.norm(best[:3, 3] - gt[:3, 3]) < 1e-3
    # score order: refinement raises the score (the starts are up to 30 degrees / 15 mm off; ICP takes nearly all of them
    # into the basin of the ground truth, so the refined scores bunch up and only this direction is a stable property)
    ctx.hypos_upload(poses)
    ctx.lcp_select_best(0.001, 10.0, 3)
    s_start = ctx.hypos_download()[1].copy()
    assert np.mean(s3 >= s_start) > 0.95 and np.median(s3) > 4 * np.median(s_start)
    # subsample against the brute-force kernels
    sub = np.arange(0, H, H // 256)[:256]
    ctx.hypos_upload(poses[sub])
    it0, cv0 = ctx.icp_refine(10, 45.0, 0.01, nn_mode=0, want_stats=True)
    p0 = ctx.hypos_download()[0].copy()
    assert (it0 == it[sub]).mean() >= 0.98 and np.array_equal(cv0, cv[sub])
    assert np.median(np.abs(p0 - refined[sub]).reshape(256, -1).max(axis=1)) < 2e-5
    ctx.hypos_upload(refined[sub])
    ctx.lcp_select_best(0.001, 10.0, 0)
    s0 = ctx.hypos_download()[1].copy()
    assert np.array_equal(s0, s2[sub])     # ordered sums: cell lists == brute force, bit for bit


def test_c5_subsample_against_the_oracle_at_size(ctx, api, orc, synth):
    """BASELINE.json configs[4] against the ORACLE (not only against the GPU's own brute-force kernels): 64 of the C5 replay
    hypotheses x the 50 000-point scene through refineByICP and computeLCP, oracle with its kd-tree (Utils.cpp:188-229, 372-444):
    ICP modes 3 (chained increments: iterations equal, poses <= 2e-5) and 6 (the reference's Levenberg-Marquardt minimiser against
    its exact-arithmetic oracle form); computeLCP ordered sums bit for bit, reduced sums <= 1e-4."""
    sc = synth.make_scene(50000, seed=13)
    mx, mn = synth.ellipsoid_model(5000)
    H = 65536
    poses = synth.replay_poses(sc.gt_pose, H, seed=13, max_rot_deg=30.0, max_trans=0.015)[:: H // 64][:64].copy()
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.0)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
    ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
    ref, rit, rcv = orc.icp_refine_batch(sc.xyz, sc.nrm, mx, mn, poses, 10, 45.0, 0.01, use_tree=True)
    ctx.hypos_upload(poses)
    it3, cv3 = ctx.icp_refine(10, 45.0, 0.01, nn_mode=3, want_stats=True)
    p3 = ctx.hypos_download()[0].copy()
    assert np.array_equal(it3, rit) and np.array_equal(cv3, rcv)
    assert np.abs(p3 - ref).max() < 2e-5
    ref6, rit6, rcv6 = orc.icp_refine_batch_lm(sc.xyz, sc.nrm, mx, mn, poses, 10, 45.0, 0.01, exact=True)
    ctx.hypos_upload(poses)
    it6, cv6 = ctx.icp_refine(10, 45.0, 0.01, nn_mode=6, want_stats=True)
    p6 = ctx.hypos_download()[0].copy()
    assert np.array_equal(cv6, rcv6) and (it6 == rit6).sum() >= 62
    d6 = np.abs(p6 - ref6).reshape(64, -1).max(axis=1)
    assert np.median(d6) < 2e-6 and np.percentile(d6, 90) < 5e-5
    for a, b in zip(p6, ref6):
        assert np.linalg.norm(a[:3, 3] - b[:3, 3]) < 1e-3 and _rot_err_deg(a[:3, :3], b[:3, :3]) < 1.0
    # (nn_mode 7 at this size: tests/test_gpu_zy_icp_canon.py::test_shipped_icp_mode_at_the_bench_sizes -- later in file order: it has never met hardware)
    sref = orc.compute_lcp_batch(sc.xyz, sc.nrm, mx, mn, ref, 0.001, 10.0, use_tree=True)
    ctx.hypos_upload(ref)
    _, _, i2 = ctx.lcp_select_best(0.001, 10.0, 2)
    assert np.array_equal(ctx.hypos_download()[1], sref) and i2 == int(np.argmax(sref))
    _, _, i3 = ctx.lcp_select_best(0.001, 10.0, 3)
    s3 = ctx.hypos_download()[1].copy()
    big = sref > 1.0
    assert np.all(np.abs(s3[big] - sref[big]) <= 1e-4 * sref[big])
    # (these replay hypotheses converge to the same pose: their scores agree to 1e-5 and the arg-max of the reduced sums may be
    # another member of that tie -- its ordered-sum score is within 1e-4 of the maximum)
    assert sref[i3] >= (1 - 1e-4) * sref.max()


def _c2_hand(ctx, api, synth, n_particles):
    import math
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    xyz, nrm = synth.make_hand_scene(hand, true, 20000, seed=5)
    swivel = xyz[xyz[:, 0] < -0.1]
    cfg = {"hand_match": {"finger1_min_match": 5, "finger2_min_match": 5, "finger1_dist_thres": 0.005, "finger2_dist_thres": 0.005,
                          "finger1_normal_angle": 60, "finger2_normal_angle": 60, "check_normal": True, "max_outter_pts": 300,
                          "outter_pt_dist": 0.002, "outter_pt_dist_weight": 1, "planar_dist_thres": 0.001,
                          "pso": {"n_pop": n_particles, "n_gen": 3, "check_freq": 10, "pso_par_c_cog": 0.1, "pso_par_c_soc": 0.9,
                                  "pso_par_initial_w": 0.0}}}
    h = api.HandT42(cfg, hand, ctx=ctx)
    h.gripper_min_dist = 0.0144
    h.setCurScene(xyz, nrm, swivel)
    return h, true, xyz, nrm, swivel


def test_c2_hand_search_at_size_against_the_oracle(ctx, api, orc, synth):
    """BASELINE.json configs[1], the hand-search half at the bench's size: 200 (+1) particles on the 20 000-point hand scene.
    objFuncPSO (Hand.cpp:10-178) on 201 angles: sum mode 0 equal to the oracle bit for bit, sum mode 1 (the bench) <= 1e-5 with
    the same arg-min; pso_int (pso.hpp:146-351) with 200 particles: the same angle and objective value as the oracle's search."""
    import ctypes as C
    from test_gpu_parity import _oracle_args
    h, true, xyz, nrm, swivel = _c2_hand(ctx, api, synth, 200)
    keep = []
    for finger, hi in (("finger_2_1", 120.0), ("finger_1_1", 120.0)):
        args = h.finger_args(finger, 0.005)
        ctx.hand_set_finger(args)
        angles = np.radians(np.linspace(0.0, hi, 201))
        oa = _oracle_args(orc, args, xyz, nrm, swivel, keep)
        ref = np.zeros(len(angles))
        orc.lib().orc_pso_objective_batch(C.byref(oa), orc.D(np.ascontiguousarray(angles)), len(angles), orc.D(ref))
        ctx.hand_set_sum_mode(0)
        got = ctx.hand_pso_eval_batch(angles)
        assert np.array_equal(got, ref), (finger, np.abs(got - ref).max())
        ctx.hand_set_sum_mode(1)
        try:
            fast = ctx.hand_pso_eval_batch(angles)
        finally:
            ctx.hand_set_sum_mode(0)
        assert np.all(np.abs(fast - ref) <= 1e-5 * np.maximum(np.abs(ref), 1.0)) and np.argmin(fast) == np.argmin(ref), finger
    # the search itself: 200 particles + centre, 3 generations
    for mode in (0, 1):
        ctx.hand_set_sum_mode(mode)
        try:
            h._tf_self["finger_1_1"] = np.eye(4, dtype=np.float32)
            assert h.matchOneComponentPSO("finger_1_1", 0, 120, False, 0.005, 60, 5)
            ang_gpu, val_gpu = h.last_angle, h.last_objval
        finally:
            ctx.hand_set_sum_mode(0)
        h._tf_self["finger_1_1"] = np.eye(4, dtype=np.float32)
        args = h.finger_args("finger_1_1", 0.005)
        oa = _oracle_args(orc, args, xyz, nrm, swivel, keep)
        s = h.pso_settings(0, 120)
        assert s.n_pop == 200
        os_ = orc.PsoSettings(s.n_pop, s.n_gen, s.check_freq, s.c_cog, s.c_soc, s.initial_w, s.w_min, s.w_max, s.err_tol, s.lower_rad, s.upper_rad, s.seed)
        ang, val = C.c_double(0), C.c_double(0)
        orc.lib().orc_pso_search(C.byref(oa), C.byref(os_), C.byref(ang), C.byref(val))
        if mode == 0:
            assert ang_gpu == np.float32(ang.value) and val_gpu == val.value
        else:   # reduced sums: the same particle wins, the objective value within 1e-5
            assert abs(float(ang_gpu) - ang.value) < 1e-6 and abs(val_gpu - val.value) <= 1e-5 * max(1.0, abs(val.value))
        assert abs(ang.value - true["finger_1_1"]) < 0.18


def test_device_resident_topk_exchange_and_frame_gather_single_rank():
    """the top-k table stays on the device from hop_topk_pack_device through ncclAllGather to the merge kernel (a
    communicator of one rank: RCCL refuses two ranks on one GPU; world 2 is covered with gloo in tests/test_distributed_cpu.py), and
    the C4 gather of per-frame poses.  Same rows as the host path (hop_topk_pack = HypoCompare order, ties by id).
    Runs in a process of its own (tests/exchange_child.py): RCCL's helper threads then live and die with that process, and the system
    ROCm it loads never meets the HIP runtime PyTorch bundles (pytest imports torch while collecting the CPU tests)."""
    import subprocess
    import sys
    env = dict(os.environ)
    if env.get("HOP_TEST_EMU"):   # (tests/emu: the child finds the model's stand-in for RCCL under the name librccl.so)
        emu_lib = env.get("HOP_TEST_EMU_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu", "_build", "libhop_emu.so")
        env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(emu_lib), "as_libhop") + os.pathsep + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "exchange_child.py")], capture_output=True, text=True, timeout=300, env=env)
    if "SKIP" in r.stdout:
        pytest.skip(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and "EXCHANGE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
