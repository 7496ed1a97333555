"""The whole driver above the C-ABI on one synthetic frame (hand + grasped object, one camera), in the call order of
src/perception/src/app/main_realdata_auto.cpp:54-205 minus the two PCL normal estimators (the frame comes with normals)
and rejectByRender: handbaseICP -> Hand::setCurScene filters -> finger PSO -> adjustHandHeight -> hand-point removal with confidences ->
generator input cloud -> runSuper4pcs -> clusterPoses -> refineByICP -> clusterPoses -> rejectByCollisionOrNonTouching ->
selectBest.  Every step runs through libhop.so; the result is judged against the frame's ground truth."""
import math
import os

import numpy as np
import pytest

from conftest import CHILD_TIMEOUT, model_time_budget_in_place

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_whole_frame_pipeline(hop):
    from hop_amd import api, config as hop_config
    from hop_amd import run_real_all as rr
    synth = hop.synth
    cfg = hop_config.load_config(os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml"))
    g = synth.grasp_frame(seed=2)
    hand = g["hand"]
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    ctx = api.Context(0)
    h = api.HandT42(cfg, hand, ctx=ctx)
    ext = np.abs(mx1.min(axis=0) - mx1.max(axis=0))
    h.gripper_min_dist = 0.8 * float(ext.min())                                      # main :41-45
    scene_xyz, scene_nrm = g["scene_xyz"], g["scene_nrm"]

    # Hand::setCurScene: handbaseICP corrects the reported hand-base pose (Hand.cpp:287, 677-777)
    handbase_in_cam, offset = h.handbaseICP(scene_xyz, scene_nrm, g["handbase_in_cam_reported"])
    err_t = np.linalg.norm(handbase_in_cam[:3, 3] - g["handbase_in_cam"][:3, 3])
    assert err_t < 2.5e-3 and h._component_status["handbase"]

    # the hand region: crop in the hand-base frame (main :79-94), 3 mm voxel grid (Hand.cpp:285), filters (Hand.cpp:289-321)
    cam_in_handbase = np.linalg.inv(handbase_in_cam.astype(np.float64))
    hb = scene_xyz.astype(np.float64) @ cam_in_handbase[:3, :3].T + cam_in_handbase[:3, 3]
    crop = (hb[:, 2] >= -0.12) & (hb[:, 2] <= 0.05) & (hb[:, 0] >= -0.25) & (hb[:, 0] <= -0.07) & (hb[:, 1] >= -0.2) & (hb[:, 1] <= 0.2)
    rx, rn = ctx.voxel_downsample_normals(scene_xyz[crop], scene_nrm[crop], 0.003)
    n_noise, n_region, n_swivel = h.setCurSceneFromRegion(rx, rn, handbase_in_cam)
    assert n_noise > 0.8 * n_region and n_swivel > 100

    # finger states (main :114-139, camera on the finger-2 side: cam_in_handbase(1,3) > 0)
    hm = cfg["hand_match"]
    assert cam_in_handbase[1, 3] > 0
    angles = {}
    for first, second in (("finger_2_1", "finger_2_2"), ("finger_1_1", "finger_1_2")):
        if h.matchOneComponentPSO(first, 0, 120, False, hm["finger1_dist_thres"], hm["finger1_normal_angle"], hm["finger1_min_match"]):
            angles[first] = h.last_angle
            if h.matchOneComponentPSO(second, 0, 90, True, hm["finger2_dist_thres"], hm["finger2_normal_angle"], hm["finger2_min_match"]):
                angles[second] = h.last_angle
    assert len(angles) == 4, angles
    for name, a in angles.items():
        assert abs(a - g["angles"][name]) < math.radians(4), (name, math.degrees(a))

    # main :141: nothing to adjust, handbaseICP already fixed the hand base (Hand.cpp:1002-1005)
    handbase_in_cam, _, counts = h.adjustHandHeight(rx, rn, handbase_in_cam)
    assert counts is None

    # hand points removed, confidences assigned (main :142-151); generator input (main :156-177)
    h.makeHandCloud()
    near = float(cfg["near_hand_dist"]) if "near_hand_dist" in cfg else 0.003
    ox, on, oc, idx = h.removeSurroundingPointsAndAssignProbability(scene_xyz, scene_nrm, handbase_in_cam, near * near)
    kept_obj = g["is_object"][idx].mean()
    assert kept_obj > 0.9 and len(ox) < len(scene_xyz)           # what remains is mostly the object
    sx, sn, sc = ctx.object_segment(ox, on, oc, 0.003)
    assert len(sx) > 300

    # pose estimation (main :183-204); as there, the estimator is constructed after the hand steps (handbaseICP used the
    # context's scene / model slots for its own ICP)
    est = api.PoseEstimator(cfg, (mx5, mn5), (mx1, mn1), ctx=ctx)
    est.setCurScene(sx, sn, sc, cloud_withouthand_raw=ox)
    est.registerHandMesh(h)
    est.registerMesh(g["object_V"], g["object_F"], "object")
    assert est.runSuper4pcs(synth.ppf_key_table())
    est.clusterPoses(30, 0.015, True)
    est.refineByICP()
    est.clusterPoses(5, 0.003, False)
    n_before = ctx.hypos_count()
    keep, diag = est.rejectByCollisionOrNonTouching(h, handbase_in_cam)
    assert 0 < keep.sum() <= n_before
    best = est.selectBest()
    gt = g["object_in_cam"].astype(np.float64)
    e = rr.adi(best._pose[:3, :3].astype(np.float64), best._pose[:3, 3].astype(np.float64), gt[:3, :3], gt[:3, 3], mx1.astype(np.float64))
    assert e < 0.005, e   # the authors' recall threshold (scripts/eval_all.py:77)


def test_whole_frame_pipeline_from_a_depth_image(hop):
    """run_real_all.cpp:100-262 from what the robot records: a 16-bit depth image, the camera matrix and the reported
    hand-base pose (synthetic grasp rendered to 640 x 480; the reference's own example frame has no matching hand or
    object model).  Nothing is injected: the normals come from the integral-image and MLS estimators, the hand region from
    the depth image, and both rejections run."""
    from hop_amd import api, config as hop_config
    from hop_amd import run_real_all as rr
    synth = hop.synth
    cfg = hop_config.load_config(os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml"))
    g = synth.grasp_depth_frame(seed=2)
    ctx = api.Context(0)
    assets = rr.Assets()
    info = {}
    pose = rr.process_frame(ctx, cfg, assets, g["depth"], g["K"], g["handbase_in_cam_reported"], info=info)
    assert info["n_valid"] > 200000 and info["n_hand_region"] > 5000
    assert np.linalg.norm(info["handbase_in_cam"][:3, 3] - g["handbase_in_cam"][:3, 3]) < 3.5e-3     # handbaseICP: 4 mm / 2 deg off -> better
    assert len(info["angles"]) == 4
    for name, a in info["angles"].items():
        assert abs(a - g["angles"][name]) < math.radians(5), (name, math.degrees(a))
    assert info["n_object_segment"] > 200 and info["n_after_icp"] >= 1
    assert 1 <= info["n_after_render"] <= info["n_after_physics"]
    gt = g["object_in_cam"].astype(np.float64)
    e = rr.adi(pose[:3, :3].astype(np.float64), pose[:3, 3].astype(np.float64), gt[:3, :3], gt[:3, 3], assets.model001[0].astype(np.float64))
    assert e < 0.005, e
    ctx.close()


def test_dataset_runner_on_the_reference_layout(hop, tmp_path):
    """run_real_all.cpp:70-273 on <base>/<model>/<record>/{rgbN.png, depthN.png, palm_in_baseN.txt, arm_left_link_7_t_N.txt},
    results in predict/<N>/model2scene.txt, evaluated like scripts/eval_all.py; a second run resumes (skips finished frames)."""
    from hop_amd import config as hop_config
    from hop_amd import run_real_all as rr
    base = str(tmp_path / "auto_collect")
    rec = rr.write_synthetic_record(base, "ellipse", n_frames=3)
    model_time_budget_in_place(os.path.join(base, "config_autodataset.yaml"))
    cfg = hop_config.load_config(os.path.join(base, "config_autodataset.yaml"))
    done = rr.run_raw(base, cfg, rank=0, world=1)
    assert done == {"synthetic_000": [0, 1, 2]}
    for k in range(3):
        assert os.path.exists(os.path.join(rec, "predict", str(k), "model2scene.txt"))
    r = rr.eval_raw(base, "ellipse", hop.synth.ellipsoid_model(4000)[0])
    assert r["total"] == 3 and r["recall_10mm"] >= 2 / 3 and r["recall_5mm"] >= 2 / 3, r["errs"]
    assert rr.run_raw(base, cfg) == {"synthetic_000": []}                       # resume: nothing left to do
    os.remove(os.path.join(rec, "predict", "1", "model2scene.txt"))
    assert rr.run_raw(base, cfg, rank=1, world=2) == {"synthetic_000": [1]}     # frame 1 belongs to rank 1 of 2


def test_cpp_drivers_from_the_depth_image_equal_the_python_mirror(hop, tmp_path):
    """The C++ host above the C-ABI from what the robot records (north_star: "host C++ calling hand-written HIP kernels ... drops in behind
    main_realdata_auto"): host/app/run_real_all (run_real_all.cpp:70-273, the reference's directory layout, 16-bit PNG depth frames and
    pose text files read in C++) and host/app/main_realdata_auto --depth (main_realdata_auto.cpp:54-205) against the Python mirror
    run_real_all.process_frame on the same frames: the same pose, hand-base correction and finger angles."""
    import subprocess
    from hop_amd import config as hop_config
    from hop_amd import run_real_all as rr
    lib = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib")
    base = str(tmp_path / "auto_collect")
    rec = rr.write_synthetic_record(base, "ellipse", n_frames=2)
    model_time_budget_in_place(os.path.join(base, "config_autodataset.yaml"))
    cfg_path = os.path.join(base, "config_autodataset.yaml")
    cfg = hop_config.load_config(cfg_path)
    assets = rr.Assets()
    adir = rr.write_assets_dir(assets, str(tmp_path / "assets"))
    # Python mirror, frame by frame
    from hop_amd import api
    ctx = api.Context(0)
    K, _, _ = rr.calibration(cfg)
    py = {}
    for idx in (0, 1):
        hb = rr.handbase_in_cam_of(cfg, rr.parse_pose_txt(os.path.join(rec, f"arm_left_link_7_t_{idx}.txt")), rr.parse_pose_txt(os.path.join(rec, f"palm_in_base{idx}.txt")))
        info = {}
        py[idx] = (rr.process_frame(ctx, cfg, assets, rr.read_depth_png(os.path.join(rec, f"depth{idx}.png")), K, hb, info=info), info, hb)
    ctx.close()
    # C++ dataset driver
    r = subprocess.run([os.path.join(lib, "run_real_all"), cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=CHILD_TIMEOUT)
    assert r.returncode == 0, r.stdout + r.stderr
    for idx in (0, 1):
        cpp = np.loadtxt(os.path.join(rec, "predict", str(idx), "model2scene.txt")).astype(np.float32)
        assert np.abs(cpp - py[idx][0]).max() < 2e-6, (idx, cpp, py[idx][0])
    r2 = subprocess.run([os.path.join(lib, "run_real_all"), cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=CHILD_TIMEOUT)
    assert r2.returncode == 0 and "0 frames written" in r2.stdout and "2 resumed" in r2.stdout            # resume
    ev = rr.eval_raw(base, "ellipse", assets.model001[0])
    assert ev["total"] == 2 and ev["recall_10mm"] >= 0.5
    # frames in flight from the C++ host (HOP_INFLIGHT workers, each with its own estimator / hand / contexts): the same files
    seq = {idx: open(os.path.join(rec, "predict", str(idx), "model2scene.txt")).read() for idx in (0, 1)}
    r4 = subprocess.run([os.path.join(lib, "run_real_all"), cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=CHILD_TIMEOUT,
                        env=dict(os.environ, HOP_INFLIGHT="2", HOP_FORCE="1"))
    assert r4.returncode == 0 and "2 frames written" in r4.stdout and "2 in flight" in r4.stdout, r4.stdout + r4.stderr
    for idx in (0, 1):
        assert open(os.path.join(rec, "predict", str(idx), "model2scene.txt")).read() == seq[idx]
    # the driver's own end-of-shard gather through RCCL (BASELINE configs[3]): HOP_GATHER_COMM=1 builds the communicator with ONE rank too, so the
    # id file, hop_comm_create, ncclAllGather inside hop_frames_allgather and the table rank 0 writes all run on this box
    if not os.environ.get("HOP_TEST_EMU"):   # (RCCL needs the device: not on the CPU model)
        env_g = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "HOP_COMM_ID_FILE")}
        r5 = subprocess.run([os.path.join(lib, "run_real_all"), cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=CHILD_TIMEOUT,
                            env=dict(env_g, HOP_GATHER="1", HOP_GATHER_COMM="1", MASTER_PORT="29411", HOP_RUN_ID="gputest"))
        assert r5.returncode == 0 and "poses of 2 frames gathered (2 from this rank, 2 rows per rank, through hop_frames_allgather)" in r5.stdout, r5.stdout + r5.stderr
        rows_all = open(os.path.join(base, "ellipse", "model2scene_all.txt")).read().strip().splitlines()
        assert [ln.split()[:2] for ln in rows_all] == [["synthetic_000", "0"], ["synthetic_000", "1"]]
        for ln, idx in zip(rows_all, (0, 1)):
            assert np.array_equal(np.array(ln.split()[2:], np.float32), np.loadtxt(os.path.join(rec, "predict", str(idx), "model2scene.txt")).astype(np.float32).reshape(16))
        assert not os.path.exists(os.path.join(base, "ellipse", ".hop_comm_id.29411.gputest"))      # rank 0 removes its id file after the collective
    # C++ single-frame driver from the depth image
    out = tmp_path / "out"
    out.mkdir()
    hbf = tmp_path / "hb.txt"
    hbf.write_text("\n".join(" ".join(repr(float(v)) for v in row) for row in py[0][2]) + "\n")
    r3 = subprocess.run([os.path.join(lib, "main_realdata_auto"), cfg_path, "--depth", adir, os.path.join(rec, "depth0.png"), str(hbf), str(out)],
                        capture_output=True, text=True, timeout=CHILD_TIMEOUT)
    assert r3.returncode == 0, r3.stdout + r3.stderr
    assert np.abs(np.loadtxt(out / "model2scene.txt").astype(np.float32) - py[0][0]).max() < 2e-6
    assert np.abs(np.loadtxt(out / "handbase_in_cam.txt").astype(np.float32) - py[0][1]["handbase_in_cam"]).max() < 2e-6
    ang = {ln.split()[0]: float(ln.split()[1]) for ln in (out / "finger_angles.txt").read_text().splitlines()}
    assert set(ang) == set(py[0][1]["angles"])
    for k, v in ang.items():
        assert abs(v - py[0][1]["angles"][k]) < 1e-6
