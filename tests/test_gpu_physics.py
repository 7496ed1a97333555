"""GPU parity of the physics row through the C-ABI: signed distances against the vectors the
reference's own libigl produced and against the oracle, the voxel grid, and the collision / non-touching decision chain
of PoseEstimator::rejectByCollisionOrNonTouching.  Run with `pytest -m gpu` on an MI355X."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
MESHES = ("ellipsoid", "box", "torus", "lshape", "single", "sheet", "degenerate", "two_parts")


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


# ------------------------------------------------------------------------------------------------ signed distance
@pytest.mark.parametrize("name", MESHES)
def test_sdf_bit_equal_to_libigl_golden(ctx, golden_dir, name):
    """Same frame: distance, sign and reported face equal libigl's (tests/golden/sdf_igl.npz) bit for bit."""
    g = np.load(os.path.join(golden_dir, "sdf_igl.npz"))
    ctx.sdf_register_mesh(5, g[f"{name}_V"], g[f"{name}_F"])
    d, f, mn, mx = ctx.sdf_signed_distance(5, g[f"{name}_P"])
    S = g[f"{name}_S"]
    assert np.array_equal(d.view(np.int32), S.view(np.int32))
    assert np.array_equal(f, g[f"{name}_I"])
    assert mn == S.min() and mx == S.max()


def test_sdf_posed_large_mesh_equals_oracle(ctx, orc, synth):
    """5120 faces registered under a pose far from the origin, 20 k queries: bit-equal to the oracle."""
    rng = np.random.default_rng(8)
    V, F = synth.ellipsoid_mesh(subdiv=4)
    T = synth.se3(synth.random_rotation(rng), [0.3, -0.2, 0.9]).astype(np.float32)
    ctx.sdf_register_mesh(6, V, F, T)
    Vt = synth.apply(T, V)
    P = np.concatenate([Vt[rng.integers(0, len(Vt), 15000)] + rng.normal(scale=0.008, size=(15000, 3)),
                        Vt.mean(0) + rng.normal(scale=0.3, size=(5000, 3))]).astype(np.float32)
    d, f, _, _ = ctx.sdf_signed_distance(6, P)
    S, I = orc.sdf_signed_distance(P, V, F, pose=T)
    assert np.array_equal(d.view(np.int32), S.view(np.int32))
    assert np.array_equal(f, I)
    assert (d < 0).sum() > 3000 and (d > 0).sum() > 3000


def test_sdf_nonconvex_sign(ctx, orc, synth):
    V, F = synth.torus_mesh(nu=48, nv=24)
    ctx.sdf_register_mesh(6, V, F)
    rng = np.random.default_rng(2)
    P = (rng.normal(size=(8000, 3)) * np.array([0.04, 0.04, 0.012])).astype(np.float32)
    d, f, _, _ = ctx.sdf_signed_distance(6, P)
    S, I = orc.sdf_signed_distance(P, V, F)
    assert np.array_equal(d.view(np.int32), S.view(np.int32)) and np.array_equal(f, I)
    exact = np.sqrt((np.linalg.norm(P[:, :2].astype(np.float64), axis=1) - 0.035) ** 2 + P[:, 2].astype(np.float64) ** 2) - 0.012
    far = np.abs(exact) > 1.5e-3  # away from the faceting error of the 1152-face torus
    assert np.array_equal(np.sign(d[far]), np.sign(exact[far]))


def test_sdf_point_on_surface_and_empty(ctx):
    V = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], np.float32)
    F = np.array([[0, 2, 1], [0, 1, 3], [0, 3, 2], [1, 2, 3]], np.int32)
    ctx.sdf_register_mesh(7, V, F)
    d, f, mn, mx = ctx.sdf_signed_distance(7, np.array([[0, 0, 0], [0.1, 0.1, 0.1], [2, 2, 2]], np.float32))
    assert np.isnan(d[0]) and f[0] == len(F) + 1  # signed_distance.cpp:150-156 with bounds of +-FLT_MAX
    assert d[1] < 0 < d[2] and abs(d[1] + 0.1) < 1e-6
    assert mn == d[1] and mx == d[2]
    d, f, mn, mx = ctx.sdf_signed_distance(7, np.zeros((0, 3), np.float32))
    assert len(d) == 0 and mn == np.finfo(np.float32).max and mx == -np.finfo(np.float32).max


def test_sdf_bad_arguments(ctx, api):
    V = np.zeros((3, 3), np.float32)
    with pytest.raises(api.HopError):
        ctx.sdf_register_mesh(99, V, np.array([[0, 1, 2]], np.int32))
    with pytest.raises(api.HopError):
        ctx.sdf_register_mesh(1, V, np.array([[0, 1, 3]], np.int32))  # vertex index out of range
    with pytest.raises(api.HopError):
        ctx.sdf_signed_distance(15, np.zeros((4, 3), np.float32))  # never registered


# ------------------------------------------------------------------------------------------------ voxel grid
def test_voxel_downsample_vs_oracle(ctx, orc):
    rng = np.random.default_rng(4)
    P = (rng.normal(size=(60000, 3)) * np.array([0.08, 0.05, 0.03]) + np.array([0.1, -0.2, 0.7])).astype(np.float32)
    P[::997] = np.nan  # non-finite points are skipped (voxel_grid.hpp:262-270)
    out = ctx.voxel_downsample(P, 0.005)
    ref = orc.voxel_downsample(P, 0.005)
    assert out.shape == ref.shape and len(out) > 5000
    # same voxels in the same order; the centroid of a voxel is a float sum whose order pcl leaves to std::sort
    assert np.abs(out - ref).max() < 1e-6
    cells_out = np.floor(out.astype(np.float32) * np.float32(200.0)).astype(np.int64)
    cells_ref = np.floor(ref.astype(np.float32) * np.float32(200.0)).astype(np.int64)
    assert (cells_out != cells_ref).any(axis=1).mean() < 0.02  # a centroid may sit on a cell boundary
    assert ctx.voxel_downsample(np.zeros((0, 3), np.float32), 0.005).shape == (0, 3)
    one = ctx.voxel_downsample(np.array([[0.1, 0.2, 0.3]], np.float32), 0.005)
    assert np.array_equal(one, np.array([[0.1, 0.2, 0.3]], np.float32))


def test_voxel_downsample_overflow_is_an_error(ctx, api):
    P = np.array([[0, 0, 0], [100, 100, 100]], np.float32)
    with pytest.raises(api.HopError):
        ctx.voxel_downsample(P, 0.001)  # 1e5^3 cells: pcl warns "integer indices would overflow" and gives up


# ------------------------------------------------------------------------------------------------ decision chain
def _register(ctx, p, posed=False):
    """Default: the pose is applied to the vertices at registration, as SDFchecker::registerMesh does (the finger meshes are
    then bit-equal to the oracle's, and so is every tie between faces).  posed: geometry registered at identity and the
    pose set afterwards (queries are moved instead)."""
    for mid, V, F, T in p["meshes"]:
        if posed:
            ctx.sdf_register_mesh(mid, V, F, None)
            ctx.sdf_set_mesh_pose(mid, T)
        else:
            ctx.sdf_register_mesh(mid, V, F, T)


def _run_case(ctx, orc, synth, p, poses, posed=False):
    _register(ctx, p, posed)
    ctx.physics_set_frame(p)
    scores = np.linspace(1.0, 0.5, len(poses)).astype(np.float32)
    ctx.hypos_upload(poses, scores)
    keep, diag = ctx.reject_by_collision()
    keep_o, diag_o = orc.reject_by_collision(p, poses)
    return keep, diag, keep_o, diag_o, scores


def _check_against_oracle(keep, diag, keep_o, diag_o):
    assert np.array_equal(diag[:, 0], diag_o[:, 0]), (diag[:, 0], diag_o[:, 0])
    assert np.array_equal(keep, keep_o)
    # the distances every check looked at: the oracle moves the mesh, the GPU moves the query (rounding of the motion)
    both = np.isfinite(diag[:, 1:]) & np.isfinite(diag_o[:, 1:])
    assert both.sum() > 0
    assert np.abs(diag[:, 1:][both] - diag_o[:, 1:][both]).max() < 2e-6
    # a check the oracle evaluated was evaluated here too
    kept = keep_o
    assert np.isfinite(diag[kept][:, 7]).all()


def test_reject_by_collision_matches_oracle(ctx, orc, synth):
    p, poses = synth.physics_case(96)
    keep, diag, keep_o, diag_o, scores = _run_case(ctx, orc, synth, p, poses)
    _check_against_oracle(keep, diag, keep_o, diag_o)
    stages = set(diag_o[:, 0].astype(int))
    assert {0, 1, 4, 5} <= stages, stages
    assert 5 < keep.sum() < len(keep)
    # the resident set is filtered in place, order and scores of the survivors preserved
    T, sc, ids = ctx.hypos_download()
    assert len(T) == keep.sum()
    assert np.array_equal(T.reshape(-1, 16), poses.reshape(-1, 16)[keep])
    assert np.array_equal(sc, scores[keep])
    assert np.array_equal(ids, np.arange(len(poses))[keep])


@pytest.mark.parametrize("status", [(1, 0, 1, 1), (0, 1, 1, 1), (1, 1, 0, 0), (0, 0, 0, 0)])
def test_reject_by_collision_component_status(ctx, orc, synth, status):
    """Fingers whose state is unknown are left out of the cloud checks (PoseEstimator.cpp:536-541, 650-652)."""
    p, poses = synth.physics_case(48, seed=5, finger_status=status)
    keep, diag, keep_o, diag_o, _ = _run_case(ctx, orc, synth, p, poses)
    _check_against_oracle(keep, diag, keep_o, diag_o)


def test_reject_by_collision_collision_stages(ctx, orc, synth):
    """A thicker object (collision threshold reached by the finger clouds and by the hand point)."""
    p, poses = synth.physics_case(64, seed=9)
    p["collision_thres"] = 0.1  # collision_dist = -7 mm: pushing the object 3 cm into a finger now collides
    p["non_touch_dist"] = 0.05
    keep, diag, keep_o, diag_o, _ = _run_case(ctx, orc, synth, p, poses)
    _check_against_oracle(keep, diag, keep_o, diag_o)
    assert {2, 3} & set(diag_o[:, 0].astype(int))


def test_reject_by_collision_tilted_fingers(ctx, orc, synth):
    """Finger links at non-zero joint angles: their meshes and clouds are no longer axis aligned in the hand-base frame
    (oriented leaf boxes and general rigid motions on every path)."""
    import math
    ang = {"finger_1_1": math.radians(14), "finger_1_2": math.radians(9), "finger_2_1": math.radians(-11), "finger_2_2": math.radians(17)}
    p, poses = synth.physics_case(96, seed=17, finger_angles=ang, mesh_subdiv=3, n_model=800)
    keep, diag, keep_o, diag_o, _ = _run_case(ctx, orc, synth, p, poses)
    _check_against_oracle(keep, diag, keep_o, diag_o)
    assert 0 < keep.sum() < len(keep)


def test_mesh_pose_equals_registration_under_the_pose(ctx, orc, synth):
    """hop_sdf_set_mesh_pose (queries moved by the inverse pose) against registering the posed vertices: same signed
    distances to the rounding of the motion, same faces; and the posed mesh answers like the oracle's posed mesh."""
    rng = np.random.default_rng(21)
    V, F = synth.torus_mesh(nu=40, nv=20)
    T = synth.se3(synth.random_rotation(rng), [0.12, -0.3, 0.8]).astype(np.float32)
    Vt = synth.apply(T, V)
    P = (Vt[rng.integers(0, len(Vt), 6000)] + rng.normal(scale=0.01, size=(6000, 3))).astype(np.float32)
    ctx.sdf_register_mesh(10, V, F, T)
    d_baked, f_baked, _, _ = ctx.sdf_signed_distance(10, P)
    ctx.sdf_register_mesh(11, V, F, None)
    ctx.sdf_set_mesh_pose(11, T)
    d_posed, f_posed, _, _ = ctx.sdf_signed_distance(11, P)
    S, I = orc.sdf_signed_distance(P, V, F, pose=T)
    assert np.array_equal(d_baked.view(np.int32), S.view(np.int32)) and np.array_equal(f_baked, I)
    assert np.abs(np.abs(d_posed) - np.abs(S)).max() < 2e-6
    clear = np.abs(S) > 5e-6
    assert np.array_equal(np.sign(d_posed[clear]), np.sign(S[clear]))
    differ = np.where(f_posed != I)[0]      # ties between faces (a closest point on a shared edge or vertex) can resolve
    assert len(differ) < 0.3 * len(I)       # differently after the motion: then the two faces share that edge or vertex
    for k in differ[:300]:
        assert len(set(F[f_posed[k]]) & set(F[I[k]])) >= 1
    ctx.sdf_set_mesh_pose(11, None)
    d_rest, _, _, _ = ctx.sdf_signed_distance(11, P)
    S0, _ = orc.sdf_signed_distance(P, V, F)
    assert np.array_equal(d_rest.view(np.int32), S0.view(np.int32))


def test_reject_by_collision_with_posed_finger_meshes(ctx, orc, synth):
    """Finger meshes registered once in their link frames and posed by hop_sdf_set_mesh_pose: the distances every check
    looks at agree with the oracle to the rounding of the motion; a decision may differ only where libigl's sign is decided
    by a tie between faces (physics rejection), which moving the query can break differently."""
    p, poses = synth.physics_case(96)
    keep, diag, keep_o, diag_o, _ = _run_case(ctx, orc, synth, p, poses, posed=True)
    same = diag[:, 0] == diag_o[:, 0]
    assert same.mean() > 0.9
    both = np.isfinite(diag[same][:, 1:7]) & np.isfinite(diag_o[same][:, 1:7])
    assert np.abs(diag[same][:, 1:7][both] - diag_o[same][:, 1:7][both]).max() < 2e-6


def test_reject_by_collision_dense_mesh_and_clouds(ctx, orc, synth):
    p, poses = synth.physics_case(24, seed=13, n_model=1500, n_scene=8000, mesh_subdiv=3, spacing=0.003)
    keep, diag, keep_o, diag_o, _ = _run_case(ctx, orc, synth, p, poses)
    _check_against_oracle(keep, diag, keep_o, diag_o)


def test_reject_requires_frame_and_handles_empty_set(api, synth):
    c = api.Context(0)
    try:
        c.hypos_upload(np.tile(np.eye(4, dtype=np.float32), (3, 1, 1)))
        with pytest.raises(api.HopError):
            c.reject_by_collision()  # no hop_physics_set_frame yet
        p, poses = synth.physics_case(8)
        for mid, V, F, T in p["meshes"]:
            c.sdf_register_mesh(mid, V, F, T)
        c.physics_set_frame(p)
        c.hypos_upload(np.zeros((0, 16), np.float32))
        keep, diag = c.reject_by_collision()
        assert len(keep) == 0 and c.hypos_count() == 0
    finally:
        c.close()


# ------------------------------------------------------------------------------------------------ host mirrors
def _write_cloud(path, xyz, nrm=None, conf=None):
    xyz = np.asarray(xyz, np.float32)
    nrm = np.zeros_like(xyz) if nrm is None else np.asarray(nrm, np.float32)
    with open(path, "wb") as f:
        np.array([len(xyz), 0 if conf is None else 1], np.int32).tofile(f)
        np.ascontiguousarray(xyz.T).tofile(f)
        np.ascontiguousarray(nrm.T).tofile(f)
        if conf is not None:
            np.asarray(conf, np.float32).tofile(f)


def _write_mesh(path, V, F):
    with open(path, "wb") as f:
        np.array([len(V), len(F)], np.int32).tofile(f)
        np.ascontiguousarray(V, np.float32).tofile(f)
        np.ascontiguousarray(F, np.int32).tofile(f)


def test_cpp_host_app_with_physics_equals_python_mirror(api, hop, synth, tmp_path):
    """main_realdata_auto.cpp:185-204 with registerHandMesh / registerMesh / rejectByCollisionOrNonTouching between the
    second clusterPoses and selectBest: the C++ host and the Python mirror keep the same hypotheses and pick the same
    pose."""
    import math
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from conftest import config_for_run
    cfg_path = config_for_run(os.path.join(root, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml"), tmp_path)
    from hop_amd import config as hop_config
    cfg = hop_config.load_config(cfg_path)
    exe = os.path.join(root, "icra20-hand-object-pose_amd", "lib", "main_realdata_auto")
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    keys = synth.ppf_key_table()
    sc = synth.make_scene(1200, seed=7)
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    hxyz, hnrm = synth.make_hand_scene(hand, true, 4000, seed=5)
    swivel = hxyz[hxyz[:, 0] < -0.1]
    # the hand holds the object: object pose in the hand-base frame as in synth.physics_case
    R_obj = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.float64)
    obj_in_hand = synth.se3(R_obj, [-0.150, 0.0, 0.0])
    handbase_in_cam = (sc.gt_pose.astype(np.float64) @ np.linalg.inv(obj_in_hand)).astype(np.float32)
    V, F = synth.ellipsoid_mesh(subdiv=3)
    frame, out = tmp_path / "frame", tmp_path / "out"
    frame.mkdir()
    out.mkdir()
    _write_cloud(frame / "model.bin", mx5, mn5)
    _write_cloud(frame / "model001.bin", mx1, mn1)
    _write_cloud(frame / "object_segment.bin", sc.xyz, sc.nrm, sc.conf)
    _write_cloud(frame / "cloud_withouthand.bin", sc.xyz, sc.nrm)
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
    with open(frame / "meshes.txt", "w") as f:
        _write_mesh(frame / "object.mesh", V, F)
        f.write("object object.mesh\n")
        for name in synth.FINGER_ORDER:
            _write_mesh(frame / f"{name}.mesh", *hand.meshes[name])
            f.write(f"{name} {name}.mesh\n")
    (frame / "handbase_in_cam.txt").write_text(" ".join(repr(float(v)) for v in handbase_in_cam.reshape(16)) + "\n")
    _write_cloud(frame / "hand_scene.bin", hxyz)
    _write_cloud(frame / "hand_region.bin", hxyz, hnrm)
    _write_cloud(frame / "hand_swivel.bin", swivel)
    (frame / "cam_side.txt").write_text("1\n")
    r = subprocess.run([exe, cfg_path, str(frame), str(out)], capture_output=True, text=True, timeout=3000)
    assert r.returncode == 0, r.stdout + r.stderr
    cpp_pose = np.loadtxt(out / "model2scene.txt").astype(np.float32)
    cpp_left = int([ln for ln in r.stdout.splitlines() if ln.startswith("hypotheses after physics:")][0].split(":")[1])

    est = api.PoseEstimator(cfg, (mx5, mn5), (mx1, mn1))
    h = api.HandT42(cfg, hand, ctx=est.ctx)
    ext = np.abs(mx1.min(axis=0) - mx1.max(axis=0))
    h.gripper_min_dist = 0.8 * float(ext.min())
    h.setCurScene(hxyz, hnrm, swivel)
    hm = cfg["hand_match"]
    for first, second in (("finger_2_1", "finger_2_2"), ("finger_1_1", "finger_1_2")):
        if h.matchOneComponentPSO(first, 0, 120, False, hm["finger1_dist_thres"], hm["finger1_normal_angle"], hm["finger1_min_match"]):
            h.matchOneComponentPSO(second, 0, 90, True, hm["finger2_dist_thres"], hm["finger2_normal_angle"], hm["finger2_min_match"])
    h.makeHandCloud()
    est.setCurScene(sc.xyz, sc.nrm, sc.conf, cloud_withouthand_raw=sc.xyz)
    est.registerHandMesh(h)
    est.registerMesh(V, F, "object")
    assert est.runSuper4pcs(keys)
    est.clusterPoses(30, 0.015, True)
    est.refineByICP()
    est.clusterPoses(5, 0.003, False)
    n_before = est.ctx.hypos_count()
    keep, diag = est.rejectByCollisionOrNonTouching(h, handbase_in_cam)
    assert len(keep) == n_before and est.ctx.hypos_count() == keep.sum() == cpp_left
    assert 0 < keep.sum()
    best = est.selectBest()
    assert np.abs(cpp_pose - best._pose).max() < 1e-6


# ------------------------------------------------------------------------------------------------ scene front end (N3b)
def test_scene_from_depth_on_the_reference_example_frame(ctx, orc, golden_dir):
    """main_realdata_auto.cpp:54-96 on the reference's example/depth7.png (tests/golden/depth7_raw.npz): 68 600 valid
    pixels -> 1 mm voxel grid -> hand-base crop; same voxels and points as the oracle."""
    g = np.load(os.path.join(golden_dir, "depth7_raw.npz"))
    lo, hi = (-0.25, -0.2, -0.12), (-0.07, 0.2, 0.05)
    xyz, counts = ctx.scene_from_depth(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    ref, counts_o = orc.scene_from_depth(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    assert counts[0] == 68600 and np.array_equal(counts, counts_o)
    assert xyz.shape == ref.shape and len(xyz) > 5000
    assert np.abs(xyz - ref).max() < 1e-6  # voxel centroids: float sums whose order pcl leaves to std::sort
    # every survivor lies in the crop box of the hand-base frame
    hb = xyz.astype(np.float64) @ g["cam_in_handbase"][:3, :3].T.astype(np.float64) + g["cam_in_handbase"][:3, 3]
    assert (hb >= np.array(lo) - 1e-5).all() and (hb <= np.array(hi) + 1e-5).all()


def test_scene_from_depth_limits_and_empty(ctx, orc):
    K = np.array([[600, 0, 320], [0, 600, 240], [0, 0, 1]], np.float32)
    I4 = np.eye(4, dtype=np.float32)
    d = np.zeros((480, 640), np.uint16)
    xyz, counts = ctx.scene_from_depth(d, 0.001, K, I4, I4, 0.001, (-1, -1, 0), (1, 1, 3))
    assert len(xyz) == 0 and counts.tolist() == [0, 0, 0]
    d[100:140, 200:260] = 100     # exactly 0.1 m: kept by every test of the chain (float 0.1f against the double literal 0.1)
    d[300:310, 300:310] = 2000    # exactly 2.0 m: `depth < 2.0` fails in convert3dOrganizedRGB
    d[10:20, 10:20] = 99
    d[400:420, 500:520] = 1500
    xyz, counts = ctx.scene_from_depth(d, 0.001, K, I4, I4, 0.005, (-1, -1, 0), (1, 1, 3))
    ref, counts_o = orc.scene_from_depth(d, 0.001, K, I4, I4, 0.005, (-1, -1, 0), (1, 1, 3))
    assert counts[0] == 40 * 60 + 20 * 20 and np.array_equal(counts, counts_o)
    assert np.abs(xyz - ref).max() < 1e-6


def test_object_segment_vs_oracle(ctx, orc, synth):
    """main_realdata_auto.cpp:156-177 on a dense synthetic cloud: same voxels, normals (towards the camera) and
    nearest-point confidences as the oracle."""
    sc = synth.make_scene(30000, seed=3)
    nrm = sc.nrm.copy()
    nrm[sc.xyz[:, 0] > np.median(sc.xyz[:, 0])] *= -1  # half of the cloud points away from the camera: the flip has work to do
    # (whole regions, not single points: opposite normals inside one voxel nearly cancel, and the sum order inside a
    # voxel is the one thing pcl leaves open)
    xyz, n2, cf = ctx.object_segment(sc.xyz, nrm, sc.conf, 0.003)
    ox, on, oc = orc.object_segment(sc.xyz, nrm, sc.conf, 0.003)
    assert xyz.shape == ox.shape and 1000 < len(xyz) < len(sc.xyz)
    assert np.abs(xyz - ox).max() < 1e-6
    assert (np.abs(n2 - on).max(axis=1) < 1e-5).mean() > 0.995  # voxels that straddle the flipped boundary may differ
    assert (cf == oc).mean() > 0.999  # a centroid that moves by 1e-7 can change its nearest dense point
    assert np.abs(np.linalg.norm(n2, axis=1) - 1).max() < 1e-5
    assert (np.einsum("ij,ij->i", n2, -xyz) >= -1e-9).all()
    e = ctx.object_segment(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32), np.zeros(0, np.float32))
    assert len(e[0]) == 0


def test_hand_scene_filters_vs_oracle(ctx, orc, synth, api, hop):
    """Hand::setCurScene (Hand.cpp:289-321): hand-base transform, radius / statistical outlier filters, swivel pass-through.
    The scene is the synthetic hand region plus isolated specks and a sparse halo; flags equal to the oracle's."""
    import math
    hand = synth.t42_hand()
    true = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    hxyz, hnrm = synth.make_hand_scene(hand, true, 6000, seed=5)
    rng = np.random.default_rng(3)
    specks = np.array([-0.15, 0.0, 0.0]) + rng.normal(scale=0.08, size=(300, 3))          # isolated: radius filters
    halo = hxyz[rng.integers(0, len(hxyz), 400)] + rng.normal(scale=0.012, size=(400, 3))  # sparse fringe: statistical filter
    xyz_h = np.concatenate([hxyz, specks, halo]).astype(np.float32)
    nrm_h = np.concatenate([hnrm, rng.normal(size=(700, 3))]).astype(np.float32)
    T = synth.se3(synth.rot_from_axis_angle([1.0, 0.2, -0.1], 2.6), [0.03, -0.02, 0.55])  # handbase_in_cam
    xyz_c, nrm_c = synth.apply(T, xyz_h), synth.rotate(T, nrm_h)
    cam_in_handbase = np.linalg.inv(T).astype(np.float32)
    hx, hn, keep, swivel = ctx.hand_scene_filters(xyz_c, nrm_c, cam_in_handbase)
    ox, on, keep_o, swivel_o = orc.hand_scene_filters(xyz_c, nrm_c, cam_in_handbase)
    assert np.array_equal(hx.view(np.int32), ox.view(np.int32)) and np.array_equal(hn.view(np.int32), on.view(np.int32))
    assert np.array_equal(keep, keep_o) and np.array_equal(swivel, swivel_o)
    assert keep[:6000].mean() > 0.9 and keep[6000:6300].mean() < 0.2 and 0 < swivel.sum() < keep.sum()
    assert keep[6300:].mean() < keep[:6000].mean()
    # mirror: the products feed the hand-state search
    from hop_amd import config as hop_config
    cfg = hop_config.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icra20-hand-object-pose_amd", "config",
                                              "config_autodataset.yaml"))
    h = api.HandT42(cfg, hand, ctx=ctx)
    h.gripper_min_dist = 0.0144
    n_noise, n_region, n_swivel = h.setCurSceneFromRegion(xyz_c, nrm_c, T)
    assert (n_noise, n_region, n_swivel) == (int(keep.sum()), len(xyz_c), int(swivel.sum()))
    hm = cfg["hand_match"]
    assert h.matchOneComponentPSO("finger_2_1", 0, 120, False, hm["finger1_dist_thres"], hm["finger1_normal_angle"], hm["finger1_min_match"])
    assert abs(h.last_angle - true["finger_2_1"]) < math.radians(6)  # 200-particle search on a 6 k-point noisy scene


def test_hand_scene_filters_small_inputs(ctx, orc):
    I4 = np.eye(4, dtype=np.float32)
    rng = np.random.default_rng(0)
    for n in (0, 1, 15, 150):
        x = (rng.normal(size=(n, 3)) * 0.005 + np.array([-0.15, 0, 0])).astype(np.float32)
        nn = rng.normal(size=(n, 3)).astype(np.float32)
        hx, hn, keep, swivel = ctx.hand_scene_filters(x, nn, I4)
        ox, on, keep_o, swivel_o = orc.hand_scene_filters(x, nn, I4)
        assert np.array_equal(keep, keep_o) and np.array_equal(swivel, swivel_o)
        assert keep.sum() == (0 if n <= 100 else keep_o.sum())  # fewer than 101 points can never pass the 0.04 m / 100 filter


def test_handbase_icp_pieces_and_mirror(ctx, orc, synth, api):
    """Hand::handbaseICP (Hand.cpp:677-777): voxel grid with normals and the source-cloud crop equal the oracle's; the
    mirror (voxel grid -> crop -> hop_icp_refine as Utils::runICP -> acceptance rules) recovers a 6 mm / 3 degree error of
    the hand-base pose and agrees with the same chain run through the oracle."""
    import math
    from hop_amd import config as hop_config
    hand = synth.t42_hand(spacing=0.004)
    rng = np.random.default_rng(11)
    pts, nrm = [], []
    for name in hand.clouds:
        Tl = np.eye(4) if name == "base_link" else synth.hand_fk(hand, {}, name)
        pts.append(synth.apply(Tl, hand.clouds[name][0]))
        nrm.append(synth.rotate(Tl, hand.clouds[name][1]))
    pts, nrm = np.concatenate(pts), np.concatenate(nrm)
    pts = (pts + rng.normal(scale=0.0003, size=pts.shape)).astype(np.float32)
    T = synth.se3(synth.rot_from_axis_angle([1.0, 0.2, -0.1], 2.6), [0.03, -0.02, 0.55])          # true handbase_in_cam
    D = synth.se3(synth.rot_from_axis_angle([0.2, 1.0, 0.3], math.radians(3.0)), [0.004, -0.003, 0.003])
    est = (T @ D).astype(np.float32)                                                              # what the robot reports
    scene_xyz, scene_nrm = synth.apply(T, pts), synth.rotate(T, nrm)
    # pieces
    vx, vn = ctx.voxel_downsample_normals(scene_xyz, scene_nrm, 0.005)
    ox, on = orc.voxel_downsample_normals(scene_xyz, scene_nrm, 0.005)
    assert vx.shape == ox.shape and np.abs(vx - ox).max() < 1e-6 and (np.abs(vn - on).max(axis=1) < 1e-5).mean() > 0.99
    cam_in_handbase = np.linalg.inv(est.astype(np.float64)).astype(np.float32)
    t1, t2 = hand.tf_in_parent["finger_1_1"], hand.tf_in_parent["finger_2_1"]
    args = (float(t1[1, 3]), float(t1[2, 3]), float(t2[1, 3]), float(t2[2, 3]))
    hx, hn, keep = ctx.handbase_region(ox, on, cam_in_handbase, *args)
    gx, gn, keep_o = orc.handbase_region(ox, on, cam_in_handbase, *args)
    assert np.array_equal(hx.view(np.int32), gx.view(np.int32)) and np.array_equal(hn.view(np.int32), gn.view(np.int32))
    assert np.array_equal(keep, keep_o) and 100 < keep.sum() < len(keep)
    # mirror
    cfg = hop_config.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icra20-hand-object-pose_amd", "config",
                                              "config_autodataset.yaml"))
    h = api.HandT42(cfg, hand, ctx=ctx)
    new, offset = h.handbaseICP(scene_xyz, scene_nrm, est)
    assert h._component_status["handbase"]
    dt = np.linalg.norm(new[:3, 3] - T[:3, 3])
    c = (np.trace(new[:3, :3].astype(np.float64).T @ T[:3, :3]) - 1) / 2
    assert dt < 1.5e-3 and math.degrees(math.acos(min(1.0, c))) < 0.6, (dt, c)
    # the same chain through the oracle
    bx, bn = hand.clouds["base_link"]
    # (Utils::runICP's minimiser is PCL's Levenberg-Marquardt: the mirror runs hop_icp_refine nn_mode 7, the oracle the same moment form)
    poses, it, cv = orc.icp_refine_batch_lm(gx[keep_o], gn[keep_o], bx, bn, np.eye(4, dtype=np.float32)[None], 50, 30.0, 0.03, moment=True)
    off_o = np.linalg.inv(poses[0].astype(np.float64))
    assert np.abs(off_o - offset).max() < 2e-4
    # a 13 degree error is found by the ICP but not accepted (rot_diff >= 10, Hand.cpp:752-756): the offset is reset
    D2 = synth.se3(synth.rot_from_axis_angle([1.0, 0.0, 0.0], math.radians(13.0)), [0.0, 0.0, 0.0])
    far = (T @ D2).astype(np.float32)
    h2 = api.HandT42(cfg, hand, ctx=ctx)
    new2, off2 = h2.handbaseICP(scene_xyz, scene_nrm, far)
    assert np.array_equal(off2, np.eye(4, dtype=np.float32)) and np.allclose(new2, far, atol=1e-6)
    assert not h2._component_status.get("handbase", False)


# ------------------------------------------------------------------------------------------------ dataset runner (N4)
def test_run_real_all_two_shards(hop, tmp_path):
    """run_real_all.cpp:70-273 / eval_all.py: eight synthetic frames processed as two shards (frame f -> rank f mod 2),
    predictions written in the reference's layout, ADI recall from the files."""
    from hop_amd import run_real_all as rr
    rec = str(tmp_path / "ellipse" / "record0")
    rr.write_synthetic_dataset(rec, 8, scene_points=1500)
    done0 = rr.run(rec, rank=0, world=2)
    assert done0 == [0, 2, 4, 6] and not os.path.exists(os.path.join(rec, "predict", "1"))
    done1 = rr.run(rec, rank=1, world=2)
    assert done1 == [1, 3, 5, 7]
    r = rr.eval_all(rec, hop.synth.ellipsoid_model(4000)[0])
    assert r["total"] == 8 and r["recall_5mm"] == 1.0 and r["recall_10mm"] == 1.0
    assert max(r["errs"].values()) < 0.002


# ------------------------------------------------------------------------------------------------ properties at size
def test_sdf_rigid_invariance_at_size(ctx, synth):
    """20 480 faces, 200 k queries (beyond what the oracle scans in seconds): a rigid motion of mesh and queries leaves the
    signed distance unchanged up to the rounding of the motion, and the sphere's analytic distance bounds the result."""
    V, F = synth.ellipsoid_mesh((0.05, 0.05, 0.05), subdiv=5)
    assert len(F) == 20480
    rng = np.random.default_rng(12)
    P = (rng.normal(size=(200000, 3)) * 0.04).astype(np.float32)
    ctx.sdf_register_mesh(8, V, F)
    d0, f0, _, _ = ctx.sdf_signed_distance(8, P)
    T = synth.se3(synth.random_rotation(rng), [0.2, -0.1, 0.7]).astype(np.float32)
    ctx.sdf_register_mesh(9, V, F, T)
    d1, f1, _, _ = ctx.sdf_signed_distance(9, synth.apply(T, P))
    ok = np.isfinite(d0) & np.isfinite(d1)
    assert ok.mean() > 0.9999
    assert np.abs(np.abs(d0[ok]) - np.abs(d1[ok])).max() < 2e-6
    clear = ok & (np.abs(d0) > 5e-6)
    assert np.array_equal(np.sign(d0[clear]), np.sign(d1[clear]))
    exact = np.linalg.norm(P.astype(np.float64), axis=1) - 0.05
    assert np.abs(d0[ok] - exact[ok]).max() < 7e-5          # faceting error of the 20 480-face sphere (sagitta of a 2.4 mm edge)
    assert (d0[ok] >= exact[ok] - 1e-6).all()               # the inscribed polyhedron is never farther inside than the sphere


def test_reject_by_collision_is_idempotent(ctx, synth):
    p, poses = synth.physics_case(512, seed=31, n_model=2000, n_scene=6000, mesh_subdiv=3, max_rot_deg=10.0, max_trans=0.006)
    _register(ctx, p)
    ctx.physics_set_frame(p)
    ctx.hypos_upload(poses)
    keep, diag = ctx.reject_by_collision()
    assert 0 < keep.sum() < len(keep)
    again, diag2 = ctx.reject_by_collision()
    assert again.all() and len(again) == keep.sum()
    assert np.array_equal(diag2[:, 1:], diag[keep][:, 1:], equal_nan=True)


def test_face_cells_with_degenerate_faces_far_from_origin(ctx, orc, synth):
    """A 340-face mesh (enough for face cells) with zero-area faces and a duplicated vertex, registered 2.5 m from the
    origin: queries inside the voxel grid, beyond it and on the grid boundary all equal the oracle bit for bit."""
    rng = np.random.default_rng(33)
    V, F = synth.ellipsoid_mesh((0.03, 0.02, 0.015), subdiv=2)
    V = np.concatenate([V, V[:1]]).astype(np.float32)
    F = np.concatenate([F, [[0, 1, 1], [5, 5, 5], [len(V) - 1, 2, 3]], F[:17][:, [0, 2, 1]]]).astype(np.int32)  # degenerate, duplicate vertex, flipped copies
    assert len(F) >= 256
    T = synth.se3(synth.random_rotation(rng), [1.5, -1.2, 1.6]).astype(np.float32)
    ctx.sdf_register_mesh(12, V, F, T)
    Vt = synth.apply(T, V)
    c = Vt.mean(0)
    P = np.concatenate([c + rng.normal(scale=0.02, size=(6000, 3)),          # inside the grid
                        c + rng.normal(scale=0.15, size=(2000, 3)),          # mostly beyond it (tree walk)
                        c + np.array([0.03 + 0.06, 0, 0]) + rng.normal(scale=1e-4, size=(500, 3))]).astype(np.float32)  # at the grid's face
    d, f, _, _ = ctx.sdf_signed_distance(12, P)
    S, I = orc.sdf_signed_distance(P, V, F, pose=T)
    assert np.array_equal(d.view(np.int32), S.view(np.int32)) and np.array_equal(f, I)


def test_adjust_hand_height_vs_oracle(ctx, orc, synth, api):
    """HandT42::adjustHandHeight (Hand.cpp:999-1051): match counts of the 13 trial heights equal the oracle's; the mirror
    recovers a hand base reported 15 mm too low."""
    from hop_amd import config as hop_config
    g = synth.grasp_frame(seed=4)
    hand = g["hand"]
    cfg = hop_config.load_config(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icra20-hand-object-pose_amd", "config",
                                              "config_autodataset.yaml"))
    h = api.HandT42(cfg, hand, ctx=ctx)
    for name, a in g["angles"].items():
        h._tf_self[name] = synth.rx(a).astype(np.float32)
    true = g["handbase_in_cam"]
    off = np.eye(4, dtype=np.float32)
    off[2, 3] = -0.015                                   # reported = true * offset^-1, so the search should answer -15 mm
    reported = (true.astype(np.float64) @ np.linalg.inv(off.astype(np.float64))).astype(np.float32)
    hand_pts = ~g["is_object"]
    rx, rn = ctx.voxel_downsample_normals(g["scene_xyz"][hand_pts], g["scene_nrm"][hand_pts], 0.003)
    new, best, counts = h.adjustHandHeight(rx, rn, reported)
    assert best == pytest.approx(-0.015) and counts.max() > 300
    assert np.abs(new - true).max() < 1e-5
    # the counts against the oracle, on the same hand-base-frame clouds
    T = np.linalg.inv(reported.astype(np.float64)).astype(np.float32)
    sx = np.stack([((T[k, 0] * rx[:, 0] + T[k, 1] * rx[:, 1]) + T[k, 2] * rx[:, 2]) + T[k, 3] for k in range(3)], axis=1).astype(np.float32)
    sn = np.stack([(T[k, 0] * rn[:, 0] + T[k, 1] * rn[:, 1]) + T[k, 2] * rn[:, 2] for k in range(3)], axis=1).astype(np.float32)
    names = sorted(h._hand_clouds)
    hx = np.concatenate([h._hand_clouds[k] for k in names])
    hn = np.concatenate([h._hand_cloud_normals[k] for k in names])
    ref = orc.hand_height_matches(sx, sn, hx, hn, h.TRIAL_HEIGHTS)
    assert np.array_equal(counts, ref)
    assert np.array_equal(ctx.hand_height_matches(sx[:0], sn[:0], hx, hn, h.TRIAL_HEIGHTS), np.zeros(13, np.int32))
    # once handbaseICP has fixed the hand base, adjustHandHeight leaves it alone (:1002-1005)
    h._component_status["handbase"] = True
    same, b2, c2 = h.adjustHandHeight(rx, rn, reported)
    assert np.array_equal(same, reported) and c2 is None


def test_registering_the_same_mesh_again_reuses_its_structures_and_a_changed_mesh_does_not(ctx, orc, synth):
    """the reference registers the object mesh once per frame (PoseEstimator.cpp:505-508); the same triangles find their tree and face cells
    standing (a fresh registration also clears hop_sdf_set_mesh_pose), one moved vertex or another pose rebuilds them"""
    import time
    rng = np.random.default_rng(5)
    V, F = synth.ellipsoid_mesh(subdiv=4)
    P = (V[rng.integers(0, len(V), 8000)] + rng.normal(scale=0.006, size=(8000, 3))).astype(np.float32)
    t0 = time.perf_counter()
    ctx.sdf_register_mesh(12, V, F)
    t_first = time.perf_counter() - t0
    d0, f0, _, _ = ctx.sdf_signed_distance(12, P)
    T = synth.se3(synth.random_rotation(rng), [0.02, 0.01, -0.03]).astype(np.float32)
    ctx.sdf_set_mesh_pose(12, T)
    t0 = time.perf_counter()
    ctx.sdf_register_mesh(12, V.copy(), F.copy())          # same content in other arrays
    t_again = time.perf_counter() - t0
    d1, f1, _, _ = ctx.sdf_signed_distance(12, P)           # ... and the pose set in between is gone, as after a fresh registration
    assert np.array_equal(d0.view(np.int32), d1.view(np.int32)) and np.array_equal(f0, f1)
    assert t_again < 0.25 * t_first
    V2 = V.copy()
    V2[17] *= 1.2                                           # one vertex moved: the structures are rebuilt
    ctx.sdf_register_mesh(12, V2, F)
    d2, f2, _, _ = ctx.sdf_signed_distance(12, P)
    S2, I2 = orc.sdf_signed_distance(P, V2, F)
    assert np.array_equal(d2.view(np.int32), S2.view(np.int32)) and np.array_equal(f2, I2) and not np.array_equal(d2, d0)
    ctx.sdf_register_mesh(12, V2, F, T)                     # the same triangles under a pose: rebuilt as well
    d3, f3, _, _ = ctx.sdf_signed_distance(12, P)
    S3, I3 = orc.sdf_signed_distance(P, V2, F, pose=T)
    assert np.array_equal(d3.view(np.int32), S3.view(np.int32)) and np.array_equal(f3, I3)
