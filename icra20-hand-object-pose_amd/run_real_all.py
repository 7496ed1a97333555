"""Dataset runner and evaluator: the frame loop of the reference's
src/perception/src/app/run_real_all.cpp:70-273 above the C-ABI, sharded over GPUs frame by frame, and the authors'
evaluator scripts/eval_all.py:11-79 with scripts/eval_utils.py:181-200 (ADI) next to it.

Directory layout = the reference's (run_real_all.cpp:72-104, scripts/eval_all.py:39-41):

    <base_dir>/<model_name>/<record>/rgb<N>.png                    colour image (only its name is used: it carries the index)
    <base_dir>/<model_name>/<record>/depth<N>.png                  16-bit depth, millimetres (Utils::readDepthImage)
    <base_dir>/<model_name>/<record>/palm_in_base<N>.txt           4 x 4, palm in the robot base (Utils::parsePoseTxt)
    <base_dir>/<model_name>/<record>/arm_left_link_7_t_<N>.txt     4 x 4, left arm link in the robot base
    <base_dir>/<model_name>/<record>/refined_gt/ob_in_cam<N>.txt   ground-truth pose, 4 x 4
    <base_dir>/<model_name>/<record>/predict/<N>/model2scene.txt   written here, read by the evaluator

`process_frame` is the body of the loop (:100-262) through the mirrors of hop_amd.api: depth image -> organised cloud ->
integral-image normals -> 1 mm voxel grid -> hand-base crop -> Hand::setCurScene (handbaseICP, outlier filters) -> finger
PSO -> adjustHandHeight -> hand-point removal with confidences -> MLS normals -> 3 mm generator cloud -> runSuper4pcs ->
clusterPoses -> refineByICP -> clusterPoses -> rejectByCollisionOrNonTouching -> rejectByRender -> selectBest.
A frame whose predict/<N>/model2scene.txt exists is skipped (resume; the reference has none).  The older prepared-cloud
layout (cloud<idx>.npz) is still read by `run`.

    python tools/run_real_all.py --base DIR [--model ellipse] [--synthetic N]                  (1 GPU)
    python -m torch.distributed.run --nproc-per-node N tools/run_real_all.py --base DIR        (one rank per GPU)

Frame idx goes to rank idx mod world; nothing is exchanged between ranks, rank 0 evaluates after a
barrier.  The datasets, meshes and URDF of the paper are not redistributed with the reference, so `write_synthetic_record`
emits frames of the synthetic grasp (stand-in T42 holding the ellipse) in this layout."""
from __future__ import annotations

import argparse
import glob
import json
import os
import re

import numpy as np
from scipy.spatial import cKDTree

from . import api, synth

SYM = {"ellipse": [180, 180, 180]}


def frame_indices(record_dir):
    out = []
    for f in glob.glob(os.path.join(record_dir, "cloud*.npz")):
        m = re.match(r"cloud(\d+)\.npz$", os.path.basename(f))
        if m:
            out.append(int(m.group(1)))
    return sorted(out)


def shard(indices, rank, world):
    """Frame f -> rank f mod world."""
    return [i for i in indices if i % world == rank]


def write_synthetic_dataset(record_dir, n_frames, scene_points=2000, seed0=1000):
    os.makedirs(os.path.join(record_dir, "refined_gt"), exist_ok=True)
    for f in range(n_frames):
        sc = synth.make_scene(scene_points, seed=seed0 + f)
        np.savez_compressed(os.path.join(record_dir, f"cloud{f}.npz"), xyz=sc.xyz, nrm=sc.nrm, conf=sc.conf)
        np.savetxt(os.path.join(record_dir, "refined_gt", f"ob_in_cam{f}.txt"), sc.gt_pose.astype(np.float64))


def estimate_frame(ctx, xyz, nrm, conf, sym):
    """main_realdata_auto.cpp:187-204 / run_real_all.cpp:225-262 without the two rejectBy* steps (they need the meshes
    of the frame's hand and object; see PoseEstimator.rejectByCollisionOrNonTouching)."""
    ctx.set_scene(xyz, nrm, conf, 0.8)
    o = ctx.default_s4pcs_opts(max_time_seconds=0)
    _, _, st = ctx.s4pcs_generate(o, download=False)
    if st.n_hypotheses == 0:
        return np.eye(4, dtype=np.float32)   # "No pose found": the reference writes the identity (main :189-196)
    ctx.cluster_poses(30.0, 0.015, sym, True)
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=api.ICP_NN_MODE_REFERENCE)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, _, _ = ctx.lcp_select_best(0.001, 10.0, -1)
    return best


def run(record_dir, model_name="ellipse", rank=0, world=1, device=0, ctx=None):
    """Processes this rank's frames; returns the indices it wrote."""
    own = ctx is None
    ctx = ctx or api.Context(device)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(4000)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(synth.ppf_key_table())
    done = []
    for idx in shard(frame_indices(record_dir), rank, world):
        g = np.load(os.path.join(record_dir, f"cloud{idx}.npz"))
        pose = estimate_frame(ctx, g["xyz"], g["nrm"], g["conf"], SYM[model_name])
        d = os.path.join(record_dir, "predict", str(idx))
        os.makedirs(d, exist_ok=True)
        np.savetxt(os.path.join(d, "model2scene.txt"), pose.astype(np.float64))
        done.append(idx)
    if own:
        ctx.close()
    return done


# ------------------------------------------------------------------------------------------------ the reference's layout
def parse_pose_txt(path):
    """Utils::parsePoseTxt (Utils.cpp:516-543): the first 16 blank-separated numbers, row-major."""
    data = []
    with open(path) as f:
        for line in f:
            data.extend(float(t) for t in line.split(" ") if t.strip())
    return np.asarray(data[:16], np.float32).reshape(4, 4)


def read_depth_png(path):
    """The 16-bit PNG behind Utils::readDepthImage (Utils.cpp:36-55); the scaling to metres happens on the GPU."""
    from PIL import Image
    a = np.asarray(Image.open(path))
    if a.dtype != np.uint16:
        a = a.astype(np.uint16)
    return np.ascontiguousarray(a)


def calibration(cfg):
    """ConfigParser.cpp:44-115: cam_K (row-major 3 x 3), cam1_in_leftarm (x y z + quaternion x y z w, normalised) and
    handbase_in_palm (row-major 4 x 4)."""
    K = np.asarray(cfg.get("cam_K", synth.CAM_K.reshape(9)), np.float32).reshape(3, 3)
    d = np.asarray(cfg.get("cam1_in_leftarm", [0, 0, 0, 0, 0, 0, 1]), np.float64)
    x, y, z, w = d[3:7] / np.linalg.norm(d[3:7])
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    cam1_in_leftarm = np.eye(4, dtype=np.float32)
    cam1_in_leftarm[:3, :3], cam1_in_leftarm[:3, 3] = R, d[:3]
    handbase_in_palm = np.asarray(cfg.get("handbase_in_palm", np.eye(4).reshape(16)), np.float32).reshape(4, 4)
    return K, cam1_in_leftarm, handbase_in_palm


def handbase_in_cam_of(cfg, leftarm_in_base, palm_in_baselink):
    """run_real_all.cpp:113-114."""
    _, cam1_in_leftarm, handbase_in_palm = calibration(cfg)
    handbase_in_leftarm = np.linalg.inv(leftarm_in_base.astype(np.float64)) @ palm_in_baselink @ handbase_in_palm
    return (np.linalg.inv(cam1_in_leftarm.astype(np.float64)) @ handbase_in_leftarm).astype(np.float32)


def raw_frame_indices(record_dir):
    """run_real_all.cpp:76-97: the index sits between "rgb" and the first "." of every file whose name contains "rgb"."""
    out = []
    for f in sorted(os.listdir(record_dir)) if os.path.isdir(record_dir) else []:
        m = re.match(r"rgb(\d+)\.", f)
        if m:
            out.append(int(m.group(1)))
    return sorted(out)


class Assets:
    """What the reference loads once per run (run_real_all.cpp:19-68): the object's clouds at 5 mm and 1 mm, its mesh, its
    PPF key table, the hand model.  The paper's files are not available: the synthetic ellipse and the stand-in T42."""

    def __init__(self, hand=None, model=None, model001=None, mesh=None, keys=None):
        self.hand = hand or synth.t42_hand()
        self.model = model or synth.ellipsoid_model_spacing(0.005)
        self.model001 = model001 or synth.ellipsoid_model(4000)
        self.mesh = mesh or synth.ellipsoid_mesh(subdiv=3)
        self.keys = keys if keys is not None else synth.ppf_key_table()


def process_frame(ctx, cfg, assets, depth_raw, K, handbase_in_cam, depth_unit=0.001, use_physics=True, use_render=True, info=None):
    """run_real_all.cpp:116-241 for one frame; returns model2scene (identity when no pose is found, main :189-196).  ``info``
    (a dict) receives the intermediate results the tests look at."""
    info = info if info is not None else {}
    ident = np.eye(4, dtype=np.float32)
    K = np.asarray(K, np.float32).reshape(3, 3)
    handbase_in_cam = np.asarray(handbase_in_cam, np.float32)
    cam_in_handbase = np.linalg.inv(handbase_in_cam.astype(np.float64)).astype(np.float32)
    # :116-151 organised cloud, integral-image normals, z pass-through, 1 mm voxel grid, hand-base crop
    organised = api.organized_cloud(depth_raw, K, depth_unit)
    org_n = ctx.normals_integral_image(organised, 0.02, 10.0, True)
    valid = ~((organised[..., 2] < 0.1) | (organised[..., 2] > 2.0))
    scene_organized, scene_organized_n = organised[valid], org_n[valid]
    scene_rgb, scene_rgb_n, counts = ctx.scene_from_depth_normals(depth_raw, depth_unit, K, cam_in_handbase, handbase_in_cam, 0.001)
    info["n_valid"], info["n_hand_region"] = int(counts[0]), len(scene_rgb)
    if len(scene_rgb) == 0:
        return ident
    # :155 Hand::setCurScene (Hand.cpp:279-334): handbaseICP on the organised cloud, 3 mm hand region, outlier filters
    h = api.HandT42(cfg, assets.hand, ctx=ctx)
    ext = np.abs(assets.model001[0].min(axis=0) - assets.model001[0].max(axis=0))
    h.gripper_min_dist = 0.8 * float(ext.min())                       # run_real_all.cpp:41-45
    fin = np.isfinite(scene_organized_n).all(axis=1)                  # (runICP drops NaN normals first, Utils.cpp:198-199)
    handbase_in_cam, offset = h.handbaseICP(scene_organized[fin], scene_organized_n[fin], handbase_in_cam)
    info["handbase_in_cam"] = handbase_in_cam
    rx, rn = ctx.voxel_downsample_normals(scene_rgb, scene_rgb_n, 0.003)
    n_noise, n_region, n_swivel = h.setCurSceneFromRegion(rx, rn, handbase_in_cam)
    info["hand_scene"] = (n_noise, n_region, n_swivel)
    # :158-185 finger states
    hm = cfg["hand_match"]
    cam_in_handbase = np.linalg.inv(handbase_in_cam.astype(np.float64))
    order = (("finger_2_1", "finger_2_2"), ("finger_1_1", "finger_1_2")) if cam_in_handbase[1, 3] > 0 else (("finger_1_1", "finger_1_2"), ("finger_2_1", "finger_2_2"))
    angles = {}
    if n_swivel > 0 and n_noise > 0:
        for first, second in order:
            if h.matchOneComponentPSO(first, 0, 120, False, hm["finger1_dist_thres"], hm["finger1_normal_angle"], hm["finger1_min_match"]):
                angles[first] = h.last_angle
                if h.matchOneComponentPSO(second, 0, 90, True, hm["finger2_dist_thres"], hm["finger2_normal_angle"], hm["finger2_min_match"]):
                    angles[second] = h.last_angle
    info["angles"] = angles
    # :187-188
    handbase_in_cam, _, _ = h.adjustHandHeight(rx, rn, handbase_in_cam)
    h.makeHandCloud()
    # :193-199 hand points removed, confidences; :201 MLS normals; :203-225 generator cloud
    near = float(cfg.get("near_hand_dist", 0.003))
    keep_n = np.isfinite(scene_rgb_n).all(axis=1)
    ox, on, oc, idx = h.removeSurroundingPointsAndAssignProbability(scene_rgb[keep_n], scene_rgb_n[keep_n], handbase_in_cam, near * near)
    info["n_without_hand"] = len(ox)
    if len(ox) < 3:
        return ident
    cloud_withouthand_raw = ox
    mp, mnrm, _, mk = ctx.normals_mls(ox, 0.003, 2)                    # object1: projected points, MLS normals, confidences kept
    sx, sn, scf = ctx.object_segment(mp, mnrm, oc[mk], 0.003)
    info["n_object_segment"] = len(sx)
    if len(sx) < 4:
        return ident
    # :230-241
    est = api.PoseEstimator(cfg, assets.model, assets.model001, ctx=ctx)
    est.setCurScene(sx, sn, scf, cloud_withouthand_raw=cloud_withouthand_raw, depth_raw=depth_raw, depth_unit=depth_unit, K=K)
    est.registerHandMesh(h)
    est.registerMesh(assets.mesh[0], assets.mesh[1], "object")
    if not est.runSuper4pcs(assets.keys):
        return ident
    est.clusterPoses(30, 0.015, True)
    est.refineByICP()
    est.clusterPoses(5, 0.003, False)
    info["n_after_icp"] = ctx.hypos_count()
    if use_physics:
        est.rejectByCollisionOrNonTouching(h, handbase_in_cam)
        info["n_after_physics"] = ctx.hypos_count()
    if use_render and ctx.hypos_count() > 0:
        est.rejectByRender(float(cfg.get("pose_estimator_wrong_ratio", 0.0)), h, handbase_in_cam, sum_mode=0)
        info["n_after_render"] = ctx.hypos_count()
    if ctx.hypos_count() == 0:
        return ident
    best = est.selectBest()
    info["score"] = best._lcp_score
    return best._pose.astype(np.float32)


def write_synthetic_record(base_dir, model_name="ellipse", record="synthetic_000", n_frames=3, seed0=2):
    """Frames of the synthetic grasp in the reference's layout, plus <base_dir>/config_autodataset.yaml: the shipped
    configuration with the synthetic camera matrix and identity robot calibration (the arm link frame is the camera frame,
    palm_in_base holds the REPORTED hand-base pose, a few millimetres / degrees off as a robot's forward kinematics are)."""
    import yaml
    from PIL import Image
    from . import config as hop_config
    rec = os.path.join(base_dir, model_name, record)
    os.makedirs(os.path.join(rec, "refined_gt"), exist_ok=True)
    cfg = hop_config.load_config()
    cfg["model_name"] = model_name
    cfg["cam_K"] = [float(v) for v in synth.CAM_K.reshape(9)]
    cfg["cam1_in_leftarm"] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0]
    cfg["handbase_in_palm"] = [float(v) for v in np.eye(4).reshape(16)]
    with open(os.path.join(base_dir, "config_autodataset.yaml"), "w") as f:
        yaml.safe_dump(cfg, f)
    fmt = lambda T: "\n".join(" ".join(repr(float(v)) for v in row) for row in np.asarray(T, np.float64)) + "\n"
    for k in range(n_frames):
        g = synth.grasp_depth_frame(seed=seed0 + k)
        Image.fromarray(g["depth"]).save(os.path.join(rec, f"depth{k}.png"))
        Image.fromarray(np.zeros((g["depth"].shape[0], g["depth"].shape[1], 3), np.uint8)).save(os.path.join(rec, f"rgb{k}.png"))
        open(os.path.join(rec, f"palm_in_base{k}.txt"), "w").write(fmt(g["handbase_in_cam_reported"]))
        open(os.path.join(rec, f"arm_left_link_7_t_{k}.txt"), "w").write(fmt(np.eye(4)))
        open(os.path.join(rec, "refined_gt", f"ob_in_cam{k}.txt"), "w").write(fmt(g["object_in_cam"]))
    return rec


def run_raw(base_dir, cfg, model_name=None, records=None, rank=0, world=1, device=0, ctx=None, assets=None, force=False, use_physics=True, use_render=True):
    """run_real_all.cpp:70-273: every frame of every record directory of <base_dir>/<model_name>/; this rank's share
    (frame index mod world); returns {record: [indices written]}."""
    own = ctx is None
    ctx = ctx or api.Context(device)
    assets = assets or Assets()
    model_name = model_name or cfg["model_name"]
    K, _, _ = calibration(cfg)
    mdir = os.path.join(base_dir, model_name)
    done = {}
    for record in (records or sorted(d for d in os.listdir(mdir) if os.path.isdir(os.path.join(mdir, d)))):
        rec = os.path.join(mdir, record)
        done[record] = []
        for idx in shard(raw_frame_indices(rec), rank, world):
            out = os.path.join(rec, "predict", str(idx), "model2scene.txt")
            if os.path.exists(out) and not force:
                continue                                            # resume: the frame was finished by an earlier run
            leftarm_in_base = parse_pose_txt(os.path.join(rec, f"arm_left_link_7_t_{idx}.txt"))
            palm_in_baselink = parse_pose_txt(os.path.join(rec, f"palm_in_base{idx}.txt"))
            depth = read_depth_png(os.path.join(rec, f"depth{idx}.png"))
            pose = process_frame(ctx, cfg, assets, depth, K, handbase_in_cam_of(cfg, leftarm_in_base, palm_in_baselink), 0.001, use_physics, use_render)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savetxt(out + ".tmp", pose.astype(np.float64))
            os.replace(out + ".tmp", out)                           # a killed run never leaves a half-written result
            done[record].append(idx)
    if own:
        ctx.close()
    return done


def eval_raw(base_dir, model_name, model_pts):
    """scripts/eval_all.py:36-79 over every record directory of one object."""
    errs = {}
    mdir = os.path.join(base_dir, model_name)
    for record in sorted(d for d in os.listdir(mdir) if os.path.isdir(os.path.join(mdir, d))):
        rec = os.path.join(mdir, record)
        for idx in raw_frame_indices(rec):
            gt_file = os.path.join(rec, "refined_gt", f"ob_in_cam{idx}.txt")
            pred_file = os.path.join(rec, "predict", str(idx), "model2scene.txt")
            pred = np.loadtxt(pred_file) if os.path.exists(pred_file) else np.eye(4)
            gt = parse_pose_txt(gt_file).astype(np.float64) if os.path.exists(gt_file) else np.eye(4)
            errs[(record, idx)] = adi(pred[:3, :3], pred[:3, 3], gt[:3, :3], gt[:3, 3], np.asarray(model_pts, np.float64))
    e = np.array(list(errs.values()))
    n = max(len(e), 1)
    return {"total": int(len(e)), "recall_5mm": float(np.sum(e < 0.005) / n), "recall_10mm": float(np.sum(e < 0.010) / n), "errs": errs}


def adi(R_est, t_est, R_gt, t_gt, pts):
    """scripts/eval_utils.py:181-200."""
    pts_est = pts @ R_est.T + t_est
    pts_gt = pts @ R_gt.T + t_gt
    nn_dists, _ = cKDTree(pts_est).query(pts_gt, k=1)
    return float(nn_dists.mean())


def eval_all(record_dir, model_pts):
    """scripts/eval_all.py:36-79 for one object: ADI of every frame that has a ground truth (a missing prediction counts
    as the identity, as there), recall at 5 mm (the authors' threshold) and at 10 mm."""
    errs = {}
    for idx in frame_indices(record_dir):
        gt_file = os.path.join(record_dir, "refined_gt", f"ob_in_cam{idx}.txt")
        pred_file = os.path.join(record_dir, "predict", str(idx), "model2scene.txt")
        pred = np.loadtxt(pred_file) if os.path.exists(pred_file) else np.eye(4)
        gt = np.loadtxt(gt_file) if os.path.exists(gt_file) else np.eye(4)
        errs[idx] = adi(pred[:3, :3], pred[:3, 3], gt[:3, :3], gt[:3, 3], np.asarray(model_pts, np.float64))
    e = np.array(list(errs.values()))
    n = max(len(e), 1)
    return {"total": int(len(e)), "recall_5mm": float(np.sum(e < 0.005) / n), "recall_10mm": float(np.sum(e < 0.010) / n), "errs": errs}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--base", default=None, help="<base_dir> of the reference's layout: <base>/<model>/<record>/{rgbN.png, depthN.png, ...}")
    ap.add_argument("--config", default=None, help="config_autodataset.yaml (default: <base>/config_autodataset.yaml if present, else the shipped one)")
    ap.add_argument("--model", default=None, help="object name (default: model_name of the config)")
    ap.add_argument("--force", action="store_true", help="recompute frames whose predict/<N>/model2scene.txt exists")
    ap.add_argument("--root", default=None, help="(older layout) record directory with cloud<idx>.npz and refined_gt/")
    ap.add_argument("--synthetic", type=int, default=0, help="write this many synthetic frames first (rank 0)")
    args = ap.parse_args()
    if not args.base and not args.root:
        ap.error("--base (the reference's layout) or --root (prepared clouds) is required")
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        local = local % max(torch.cuda.device_count(), 1)  # self-test: several ranks on one GPU (HOP_BENCH_BACKEND=gloo)
        torch.cuda.set_device(local)
        dist.init_process_group(backend=os.environ.get("HOP_BENCH_BACKEND", "nccl"))
    if args.base:
        from . import config as hop_config
        if args.synthetic and rank == 0:
            write_synthetic_record(args.base, args.model or "ellipse", n_frames=args.synthetic)
        if dist is not None:
            dist.barrier()
        cfg_path = args.config or (os.path.join(args.base, "config_autodataset.yaml") if os.path.exists(os.path.join(args.base, "config_autodataset.yaml")) else None)
        cfg = hop_config.load_config(cfg_path)
        done = run_raw(args.base, cfg, args.model, rank=rank, world=world, device=local, force=args.force)
        if dist is not None:
            dist.barrier()
        if rank == 0:
            r = eval_raw(args.base, args.model or cfg["model_name"], synth.ellipsoid_model(4000)[0])
            r.pop("errs")
            print(json.dumps({"frames_this_rank": sum(len(v) for v in done.values()), "world": world, **r}))
    else:
        if args.synthetic and rank == 0:
            write_synthetic_dataset(args.root, args.synthetic)
        if dist is not None:
            dist.barrier()
        done = run(args.root, rank=rank, world=world, device=local)
        if dist is not None:
            dist.barrier()
        if rank == 0:
            r = eval_all(args.root, synth.ellipsoid_model(4000)[0])
            r.pop("errs")
            print(json.dumps({"frames_this_rank": len(done), "world": world, **r}))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
