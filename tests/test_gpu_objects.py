"""BASELINE configs[2] ("cuboid + cylinder + tless batch ...") on stand-ins -- `pytest -m gpu` on an MI355X.

The reference's meshes and the 986 real frames are download links (README.md:49-54), so the other objects of
config_autodataset.yaml:35-59 run on synthetic stand-ins with the SAME symmetry classes (hop_amd.synth.OBJECT_SYMMETRY): cuboid
x/y 180, z 90; cylinder x/y 180, z 0 (continuous); tless3 y 0, x/z 360; mustard 360 (none) -- the values clusterPoses folds the Euler
differences with (PoseEstimator.cpp:135-196).  Per object:
  * the offline tool computePPF.cpp:56-107 (1 mm grid, centring, MLS normals r = 3 mm, outward flip, 5 mm grid, pair keys) chained from
    the library's calls (api.compute_ppf) against the same chain through the oracle;
  * the as-shipped chain generate -> clusterPoses(30, 15 mm) -> refineByICP -> clusterPoses(5, 3 mm) -> selectBest with the object's
    symmetry, GPU against oracle (same selected pose within 1 mm / 1 degree) and against the ground truth (ADI < 5 mm, the authors'
    threshold, scripts/eval_all.py:77).
"""
import math

import numpy as np
import pytest
from scipy.spatial import cKDTree

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


def _rot_err_deg(Ra, Rb):
    c = (np.trace(Ra.T.astype(np.float64) @ Rb.astype(np.float64)) - 1) / 2
    return math.degrees(math.acos(max(-1.0, min(1.0, float(c)))))


def _oracle_compute_ppf(orc, cloud):
    x1 = orc.voxel_downsample(cloud, 0.001)
    mid = ((x1.min(axis=0) + x1.max(axis=0)).astype(np.float32).astype(np.float64) / 2.0).astype(np.float32)
    xc = (x1 - mid).astype(np.float32)
    px, pn, _, _ = orc.normals_mls(xc, 0.003, 2)
    flip = np.einsum("ij,ij->i", -px, pn) < 0
    pn = (-np.where(flip[:, None], -pn, pn)).astype(np.float32)
    m5x, m5n = orc.voxel_downsample_normals(px, pn, 0.005)
    return dict(model001=(px, pn), model=((m5x + mid).astype(np.float32), m5n), keys=orc.model_ppf_keys(m5x, m5n), mid=mid)


@pytest.mark.parametrize("name", ["cuboid", "cylinder", "tless3"])
def test_compute_ppf_tool_chain_equals_oracle(ctx, api, orc, hop, name):
    synth = hop.synth
    rng = np.random.default_rng(5)
    dense, _ = synth.object_surface(name, 0.0007)
    dense = (dense + np.float32([0.01, -0.02, 0.03]) + rng.normal(scale=2e-5, size=dense.shape)).astype(np.float32)   # off-centre, like a scanned model
    g = api.compute_ppf(ctx, dense)
    o = _oracle_compute_ppf(orc, dense)
    assert np.array_equal(g["mid"], o["mid"])
    assert g["model001"][0].shape == o["model001"][0].shape and len(g["model001"][0]) > 5000
    assert np.abs(g["model001"][0] - o["model001"][0]).max() < 2e-6
    assert np.abs(g["model001"][1] - o["model001"][1]).max() < 5e-5
    # the normals point outward: away from the centred origin for these star-shaped stand-ins (tless3: nearly everywhere)
    outward = np.einsum("ij,ij->i", g["model001"][0], g["model001"][1]) > 0
    assert outward.mean() > (0.93 if name == "tless3" else 0.999)
    assert g["model"][0].shape == o["model"][0].shape
    assert np.abs(g["model"][0] - o["model"][0]).max() < 2e-6 and np.abs(g["model"][1] - o["model"][1]).max() < 5e-5
    # key tables: the pair loop on identical clouds is exact (tests/test_gpu_parity.py); through the chain, a normal that differs by
    # 1e-5 moves a key across a 10 degree bin edge now and then
    kg, ko = set(map(tuple, g["keys"].tolist())), set(map(tuple, o["keys"].tolist()))
    assert len(kg & ko) >= 0.995 * len(kg | ko) and len(kg) > 1000
    assert np.array_equal(ctx.model_ppf_keys(*[np.ascontiguousarray(a - (o["mid"] if k == 0 else 0)) for k, a in enumerate(o["model"])]), o["keys"])


def _min_sym_rot_err(synth, name, R, Rgt, steps=36):
    return min(_rot_err_deg(R, Rgt @ S) for S in synth.symmetry_rotations(name, steps))


@pytest.mark.parametrize("name", OBJECTS)
def test_as_shipped_chain_per_object_against_oracle_and_ground_truth(ctx, api, orc, hop, name):
    synth = hop.synth
    sym = list(synth.OBJECT_SYMMETRY[name])
    mx5, mn5 = synth.object_model(name, 0.005)
    mx1, mn1 = synth.object_model(name, 0.0015)
    sc = synth.make_object_scene(name, 1500, seed=31)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    keys = ctx.model_ppf_keys(mx5, mn5)
    assert np.array_equal(keys, orc.model_ppf_keys(mx5, mn5))
    ctx.set_ppf_keys(keys)
    o = ctx.default_s4pcs_opts(max_time_seconds=0)
    pose, lcp, st = ctx.s4pcs_generate(o)
    assert len(lcp) > 50
    ctx.cluster_poses(30.0, 0.015, sym, True)
    n1 = ctx.hypos_count()
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=api.ICP_NN_MODE_REFERENCE)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, 2)
    # oracle chain
    oo = orc.OracleS4PCS()
    oo.set_keys(keys)
    oo.run(sc.xyz, sc.nrm, sc.conf, mx5, mn5, 1)
    op, ol = oo.hypos()
    assert np.array_equal(ol, lcp) and np.array_equal(op[:, :3, :3], pose[:, :3, :3])
    k1 = orc.cluster_poses(op, ol, np.arange(len(ol)), 30.0, 0.015, sym)
    assert len(k1) == n1          # the symmetry folding keeps the same cluster heads
    p1, l1 = op[k1][:100], ol[k1][:100]
    p2, _, _ = orc.icp_refine_batch_lm(sc.xyz, sc.nrm, mx5, mn5, p1, 10, 45.0, 0.01, moment=True)   # (the mirrors' nn_mode 7: same bits, tests/test_gpu_zy_icp_canon.py)
    k2 = orc.cluster_poses(p2, l1, np.arange(len(l1)), 5.0, 0.003, sym)
    p3 = p2[k2]
    s3 = orc.compute_lcp_batch(sc.xyz, sc.nrm, mx1, mn1, p3, 0.001, 10.0)
    ob = p3[int(np.flatnonzero(s3 == s3.max())[0])]
    # the same pose within 1 mm / 1 degree -- modulo the object's symmetry group: among the equivalent cluster heads of a continuous
    # symmetry (cylinder z, tless3 y) the arg-max of near-equal scores may pick another member (sampled every 0.5 degree here)
    assert np.linalg.norm(best[:3, 3] - ob[:3, 3]) < 1e-3
    assert _min_sym_rot_err(synth, name, best[:3, :3].astype(np.float64), ob[:3, :3].astype(np.float64), 720) < 1.0
    # (equivalent heads of a continuous symmetry score within 1e-3 of each other, and so do the two chains' picks)
    assert abs(score - s3.max()) <= (2e-3 if min(sym) == 0 else 1e-4) * s3.max()
    # ground truth: ADI (scripts/eval_utils.py:181-200) below the authors' 5 mm, rotation right modulo the object's symmetry group
    a = mx1.astype(np.float64) @ best[:3, :3].T.astype(np.float64) + best[:3, 3]
    b = mx1.astype(np.float64) @ sc.gt_pose[:3, :3].T + sc.gt_pose[:3, 3]
    assert cKDTree(a).query(b)[0].mean() < 0.005
    assert _min_sym_rot_err(synth, name, best[:3, :3].astype(np.float64), sc.gt_pose[:3, :3]) < 6.0
