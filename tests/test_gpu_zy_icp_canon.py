"""Row S2 in the form the mirrors ship (hop_icp_refine nn_mode 7: the reference's minimiser from integer-exact moment sums, IEEE-only solve)
-- `pytest -m gpu` on an MI355X.

VERDICT r03's largest open parity item: with nn_mode 6 the GPU chain and the CPU chain returned different poses on 10-30 % of the frames
of every object but the ellipse -- float sums added in the order of the kernel's tiles, reciprocal estimates in the solve, and Eigen's
float forward differences turn 1e-16 into 1e-3.  nn_mode 7 removes the cause instead of bounding the effect: the moment sums are exact
integers (any order gives the same sum) and every operation after them is an IEEE operation in a fixed order, so the oracle's statement of
the algorithm (minimiser 7; pinned to Eigen's own run by tests/test_lm_core_cpu.py, tests/test_icp_lm_oracle.py) returns THE SAME BITS.
Asserted here:
  * refined poses, iteration counts and convergence flags of hop_icp_refine(nn_mode 7) equal the oracle's bit for bit -- synthetic sets up
    to the C2 size, the C1 frame (example/depth7.png hand region), hypotheses that do not converge;
  * the as-shipped chain generate -> clusterPoses -> refineByICP -> clusterPoses -> selectBest returns the oracle chain's pose BIT FOR
    BIT (the north star's criterion is 1 mm / 1 degree) for every stand-in object over ten frames each, on the C1 frame and on the
    ellipse at 1 200 points;
  * against Eigen's own float run (golden vectors): as close to the reference's default build as its -march=native build is.
nn_mode 5 / 6 (the float-faithful pin and the fast float moment form) keep their own tests in tests/test_gpu_icp_lm.py.
(Late in file order on purpose -- "zy": after every file whose kernels have run on hardware, before the variants and the instruction
self-test: `pytest -x` must not stop on code that has never met a device before the rest was seen.)
"""
import math
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

from test_icp_lm_oracle import assert_as_close_as_the_other_build, c1_inputs

pytestmark = pytest.mark.gpu
OBJECTS = ["cuboid", "cylinder", "tless3", "mustard", "ellipse"]


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


def biteq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))


def _gpu(ctx, api, xyz, nrm, conf, mx5, mn5, poses, max_hypotheses=0, nn_mode=7):
    ctx.set_scene(xyz, nrm, conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.hypos_upload(poses)
    it, cv = ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=max_hypotheses, nn_mode=nn_mode, want_stats=True)
    p, _, _ = ctx.hypos_download()
    return p, it, cv


@pytest.mark.parametrize("ns,nh,rot,trans,seed", [(1500, 64, 10.0, 0.005, 1003), (4000, 128, 25.0, 0.012, 7), (20000, 256, 30.0, 0.015, 7), (777, 33, 40.0, 0.02, 5)])
def test_nn_mode7_returns_the_oracles_bits(ctx, api, orc, hop, ns, nh, rot, trans, seed):
    synth = hop.synth
    sc = synth.make_scene(ns, seed=seed)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, nh, seed=4, max_rot_deg=rot, max_trans=trans)
    keep = sc.conf >= 0.8
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx5, mn5, poses, 10, 45.0, 0.01, moment=True)
    pg, itg, cvg = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    assert np.array_equal(cvg, cvo) and np.array_equal(itg, ito)
    assert biteq(pg, po), float(np.abs(pg - po).max())
    # a second run of the same call, and the same hypotheses in another order: the same bits (no dependence on the launch)
    pg2, itg2, _ = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    assert biteq(pg2, pg) and np.array_equal(itg2, itg)
    o = np.random.default_rng(1).permutation(nh)
    pg3, itg3, _ = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses[o])
    assert biteq(pg3, pg[o]) and np.array_equal(itg3, itg[o])


def test_nn_mode7_scene_order_does_not_matter(ctx, api, orc, hop):
    """the scene handed over in another order lands in other lanes / wavefronts / workgroups: integer sums, same bits"""
    synth = hop.synth
    sc = synth.make_scene(3000, seed=12)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, 48, seed=9, max_rot_deg=20.0, max_trans=0.01)
    pa, ita, cva = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    o = np.random.default_rng(3).permutation(len(sc.xyz))
    pb, itb, cvb = _gpu(ctx, api, sc.xyz[o], sc.nrm[o], sc.conf[o], mx5, mn5, poses)
    assert biteq(pa, pb) and np.array_equal(ita, itb) and np.array_equal(cva, cvb)


def test_nn_mode7_not_converged_and_few_correspondences(ctx, api, orc, hop):
    """hypotheses far from the scene (no correspondences -> not converged -> identity, Utils.cpp:218-225) next to good ones"""
    synth = hop.synth
    sc = synth.make_scene(600, seed=5)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    poses = synth.replay_poses(sc.gt_pose, 12, seed=2, max_rot_deg=5.0, max_trans=0.003)
    poses[3, :3, 3] += 0.5
    poses[7, :3, 3] -= 0.3
    keep = sc.conf >= 0.8
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx5, mn5, poses, 10, 45.0, 0.01, moment=True)
    pg, itg, cvg = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, poses)
    assert cvo[3] == 0 and cvo[7] == 0 and np.array_equal(cvg, cvo) and np.array_equal(itg, ito)
    assert np.array_equal(pg[3], poses[3]) and np.array_equal(pg[7], poses[7])
    assert biteq(pg, po)


def test_nn_mode7_c1_depth7_bits_and_distance_to_eigens_run(ctx, api, orc, hop, golden_dir):
    """BASELINE configs[0]: the hand region of the reference's example/depth7.png, refineByICP's <= 100 hypotheses -- the oracle's bits, and
    as close to Eigen's own float run (tests/golden/icp_lm_c1.npz, made from the reference's vendored Eigen) as the reference's other
    build is.  (On this frame -- a hand, no ellipse: every hypothesis a poor local optimum -- two builds of the reference agree within
    1 mm / 1 degree on 21 of 100 hypotheses only: the SELECTED pose of C1 is not reproducible by the reference itself, which is why the
    chain criterion below is stated against the oracle and the distance to Eigen's run as a comparison with the reference's own spread.)"""
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    xyz, nrm, mx5, mn5 = c1_inputs(hop, golden_dir)
    conf = np.ones(len(xyz), np.float32)
    p, it, cv = _gpu(ctx, api, xyz, nrm, conf, mx5, mn5, g["poses_in"], max_hypotheses=100)
    po, ito, cvo = orc.icp_refine_batch_lm(xyz, nrm, mx5, mn5, g["poses_in"][:100], 10, 45.0, 0.01, moment=True)
    assert biteq(p, po) and np.array_equal(it, ito) and np.array_equal(cv, cvo)
    assert_as_close_as_the_other_build(p, g, slack=5)
    assert (cv == g["converged"]).sum() >= 88
    g2 = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    sc = hop.synth.make_scene(4000, seed=7)
    p2, it2, cv2 = _gpu(ctx, api, sc.xyz, sc.nrm, sc.conf, mx5, mn5, g2["poses_in"])
    mine, native = assert_as_close_as_the_other_build(p2, g2, slack=3)
    assert mine[0] >= 88 and (it2 == g2["iterations"]).sum() >= 88 and np.array_equal(cv2, g2["converged"])


def _chains(ctx, api, orc, xyz, nrm, conf, mx5, mn5, mx1, mn1, keys, sym):
    """the as-shipped chain (main_realdata_auto.cpp:187-204 minus the two rejectBy* rows) on the GPU and in the oracle"""
    ctx.set_scene(xyz, nrm, conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(keys)
    pose, lcp, st = ctx.s4pcs_generate(ctx.default_s4pcs_opts(max_time_seconds=0))
    ctx.cluster_poses(30.0, 0.015, sym, True)
    n1 = ctx.hypos_count()
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=7)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, 2)
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(xyz, nrm, conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert np.array_equal(ol, lcp) and biteq(op[:, :3, :3], pose[:, :3, :3])
    k1 = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, sym)
    assert len(k1) == n1
    keep = conf >= 0.8
    p1, l1 = op[k1][:100], ol[k1][:100]
    p2, _, _ = orc.icp_refine_batch_lm(xyz[keep], nrm[keep], mx5, mn5, p1, 10, 45.0, 0.01, moment=True)
    k2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, sym)
    p3 = p2[k2]
    s3 = orc.compute_lcp_batch(xyz[keep], nrm[keep], mx1, mn1, p3, 0.001, 10.0)
    ob = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    return best, score, ob, float(s3.max())


@pytest.mark.parametrize("seed", list(range(31, 41)))
@pytest.mark.parametrize("name", OBJECTS)
def test_as_shipped_chain_returns_the_oracle_chains_pose(ctx, api, orc, hop, name, seed):
    """BASELINE configs[2] on the stand-in objects (the reference's symmetry classes), ten frames each: the pose the GPU chain returns IS the
    pose the CPU chain returns (north star: within 1 mm / 1 degree), and it is a correct pose (ADI below the authors' 5 mm)."""
    synth = hop.synth
    sym = list(synth.OBJECT_SYMMETRY[name])
    mx5, mn5 = synth.object_model(name, 0.005)
    mx1, mn1 = synth.object_model(name, 0.0015)
    sc = synth.make_object_scene(name, 1500, seed=seed)
    keys = orc.model_ppf_keys(mx5, mn5)
    best, score, ob, so = _chains(ctx, api, orc, sc.xyz, sc.nrm, sc.conf, mx5, mn5, mx1, mn1, keys, sym)
    assert biteq(best, ob), (float(np.linalg.norm(best[:3, 3] - ob[:3, 3]) * 1e3), "mm")
    assert np.float32(score) == np.float32(so)
    if seed == 31:
        a = mx1.astype(np.float64) @ best[:3, :3].T.astype(np.float64) + best[:3, 3]
        b = mx1.astype(np.float64) @ sc.gt_pose[:3, :3].T + sc.gt_pose[:3, 3]
        assert cKDTree(a).query(b)[0].mean() < 0.005


def test_c1_depth7_chain_returns_the_oracle_chains_pose(ctx, api, orc, hop, golden_dir):
    """BASELINE configs[0] with the minimiser the reference runs: as-shipped options on the hand-region cloud of example/depth7.png"""
    synth = hop.synth
    g = np.load(os.path.join(golden_dir, "depth7_hand_region.npz"))
    xyz, nrm = g["xyz"], g["nrm"]
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    best, score, ob, so = _chains(ctx, api, orc, xyz, nrm, np.ones(len(xyz), np.float32), mx5, mn5, mx1, mn1, synth.ppf_key_table(), [180, 180, 180])
    assert biteq(best, ob) and np.float32(score) == np.float32(so)


def test_ellipse_chain_returns_the_oracle_chains_pose_and_the_truth(ctx, api, orc, hop):
    synth = hop.synth
    sc = synth.make_scene(1200, seed=7)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    best, score, ob, so = _chains(ctx, api, orc, sc.xyz, sc.nrm, sc.conf, mx5, mn5, mx1, mn1, synth.ppf_key_table(), [180, 180, 180])
    assert biteq(best, ob) and np.float32(score) == np.float32(so)
    a = mx1 @ best[:3, :3].T + best[:3, 3]
    b = mx1 @ sc.gt_pose[:3, :3].T + sc.gt_pose[:3, 3]
    assert cKDTree(b).query(a)[0].mean() < 0.005


@pytest.mark.parametrize("config", ["C2", "C5"])
def test_shipped_icp_mode_at_the_bench_sizes(ctx, api, orc, hop, config, monkeypatch):
    """nn_mode 7 (what the mirrors and the bench run) at the sizes the bench runs it at -- C2: 10 240 hypotheses x the
    20 000-point scene, C5: one GPU's share, 8 192 x 50 000.  The WHOLE set: two runs return the same bits, the same hypotheses in another
    order return the same bits (integer sums: no dependence on the launch), hypothesis batches of another size (HOP_ICP_WS_CAP_MB: h0 > 0)
    return the same bits; the ORACLE's bits on a 256-hypothesis subsample spread over the whole set.  (On the CPU model of tests/emu the
    set is 384 hypotheses: the sizes are the point of this test on a device only.)"""
    synth = hop.synth
    emu = bool(os.environ.get("HOP_TEST_EMU"))
    ns, H, seed = (20000, 10240, 7) if config == "C2" else (50000, 8192, 13)
    if emu:
        ns, H = ns // 10, 384
    sc = synth.make_scene(ns, seed=seed)
    mx, mn = synth.ellipsoid_model(5000)
    thr = 0.8 if config == "C2" else 0.0
    poses = synth.replay_poses(sc.gt_pose, H, seed=seed, max_rot_deg=30.0, max_trans=0.015)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, thr)
    ctx.set_model(api.HOP_MODEL_5MM, mx, mn)

    def run(p):
        ctx.hypos_upload(p)
        it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=7, want_stats=True)
        return ctx.hypos_download()[0].copy(), it.copy(), cv.copy()
    p1, it1, cv1 = run(poses)
    assert cv1.mean() > 0.9 and it1.max() <= 10 and np.isfinite(p1).all()
    p2, it2, cv2 = run(poses)
    assert np.array_equal(p1.view(np.int32), p2.view(np.int32)) and np.array_equal(it1, it2) and np.array_equal(cv1, cv2)
    o = np.random.default_rng(3).permutation(H)
    p3, it3, cv3 = run(poses[o])
    assert np.array_equal(p3.view(np.int32), p1[o].view(np.int32)) and np.array_equal(it3, it1[o]) and np.array_equal(cv3, cv1[o])
    # batches of ~H / 5 hypotheses (4 bytes per scene point and hypothesis of workspace): the state of batch h0 > 0, the last batch short
    monkeypatch.setenv("HOP_ICP_WS_CAP_MB", str(max(1, (4 * ns * H // 5) >> 20)))
    p4, it4, cv4 = run(poses)
    monkeypatch.delenv("HOP_ICP_WS_CAP_MB")
    assert np.array_equal(p4.view(np.int32), p1.view(np.int32)) and np.array_equal(it4, it1) and np.array_equal(cv4, cv1)
    # the oracle (kd-tree NN, minimiser 7) on a subsample drawn across the whole set
    sub = np.unique(np.linspace(0, H - 1, 64 if emu else 256).astype(int))
    keep = sc.conf >= thr
    po, ito, cvo = orc.icp_refine_batch_lm(sc.xyz[keep], sc.nrm[keep], mx, mn, np.ascontiguousarray(poses[sub]), 10, 45.0, 0.01, moment=True)
    assert np.array_equal(it1[sub], ito) and np.array_equal(cv1[sub], cvo)
    assert np.array_equal(p1[sub].view(np.int32), np.ascontiguousarray(po, np.float32).view(np.int32))


def test_stage_min_switch_has_an_effect(api, hop, orc):
    """hop_ctx_h2d / hop_ctx_d2h called themselves below HOP_STAGE_MIN; the branch now hands small copies to the runtime
    directly.  A child process with HOP_STAGE_MIN set far above every transfer runs a small ICP and returns the same bits."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "\n".join([
        "import os, sys, numpy as np",
        "sys.path.insert(0, %r)" % root,
        "import hop_loader; hop = hop_loader.load()",
        "from hop_amd import api",
        "if os.environ.get('HOP_TEST_EMU'):   # (tests/emu: the same child on the CPU model)",
        "    api.LIB_PATH = %r; api._lib = None" % os.path.join(root, "tests", "emu", "_build", "libhop_emu.so"),
        "s = hop.synth; sc = s.make_scene(500, seed=3); mx, mn = s.ellipsoid_model_spacing(0.005)",
        "c = api.Context(0); c.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8); c.set_model(api.HOP_MODEL_5MM, mx, mn)",
        "c.hypos_upload(s.replay_poses(sc.gt_pose, 9, seed=1, max_rot_deg=8.0, max_trans=0.004))",
        "c.icp_refine(10, 45.0, 0.01, nn_mode=7); p = c.hypos_download()[0]",
        "print(p.view(np.int32).sum(dtype=np.int64))"])
    env = dict(os.environ)
    outs = []
    for v in (None, "1000000000", "1"):
        e = dict(env)
        e.pop("HOP_STAGE_MIN", None)
        if v:
            e["HOP_STAGE_MIN"] = v
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2]


def test_device_sort_and_host_merge_order_scores_the_same_way(ctx, api):
    """the device sorted (score, id) rows by a bit-pattern key, the host merge by float comparison; they disagreed on -0 vs +0 and
    on NaN.  One key now (score_order_key, csrc/hop_math.h: -0 = +0, NaN below every number): hop_topk_pack (k_score_keys + radix sort on the
    device) and hop_topk_merge (host) return the same order for the same rows."""
    scores = np.array([0.0, -0.0, np.nan, 1.5, -2.0, -np.inf, 1.5, 0.25], np.float32)
    poses = np.tile(np.eye(4, dtype=np.float32), (len(scores), 1, 1))
    poses[:, 0, 3] = np.arange(len(scores))            # (tells the rows apart)
    ctx.hypos_upload(poses, scores)
    rows, n = ctx.topk_pack(len(scores))
    assert n == len(scores)
    dev_ids = rows[:, 1].copy().view(np.int32).tolist()
    assert dev_ids == [3, 6, 7, 0, 1, 4, 5, 2]          # 1.5 (ids 3, 6) | 0.25 | +-0 by id | -2 | -inf | NaN last
    merged, m = api.topk_merge(rows[None], len(scores))
    assert m == len(scores) and merged.reshape(len(scores), -1)[:, 1].copy().view(np.int32).tolist() == dev_ids
