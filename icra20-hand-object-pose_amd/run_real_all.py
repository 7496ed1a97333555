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
    ctx.icp_refine_reference(10, 45.0, 0.01, max_hypotheses=100)
    ctx.cluster_poses(5.0, 0.003, sym, False)
    best, _, _ = ctx.lcp_select_best(0.001, 10.0, -1)
    return best


def run(record_dir, model_name="ellipse", rank=0, world=1, device=0, ctx=None, poses_out=None):
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
        if poses_out is not None:
            poses_out[idx] = pose
    if own:
        ctx.close()
    return done


def gather_frame_poses(local, all_keys, rank, world, comm=None, dist=None):
    """BASELINE configs[3] "RCCL gather of per-frame best pose" / SURVEY 8(e) "C4 frames": every rank holds the poses of the frames it
    processed (`local`: key -> 4 x 4); `all_keys` is the globally agreed, sorted list of frame keys (every rank enumerates the same
    directory).  ONE collective at the end of the shard: rows {frame number, pose} padded to the largest shard --
    hop_frames_allgather (ncclAllGather inside libhop.so) when an RCCL communicator is given, torch.distributed.all_gather (gloo: the
    CPU tests) otherwise.  Returns {key: pose} of ALL frames on every rank."""
    number = {k: i for i, k in enumerate(all_keys)}
    # the largest shard, which every rank derives from the same list and the same rule (frame index mod world, per record)
    owner = lambda k: (k[1] if isinstance(k, tuple) else k) % world
    counts = [0] * world
    for k in all_keys:
        counts[owner(k)] += 1
    rows_per_rank = max(1, max(counts))
    rows = np.zeros((len(local), api.FRAME_ROW_FLOATS), np.float32)
    for r, (k, pose) in enumerate(sorted(local.items(), key=lambda kv: number[kv[0]])):
        rows[r, 0] = number[k]
        rows[r, 1:] = np.asarray(pose, np.float32).reshape(16)
    if len(rows) > rows_per_rank:
        raise ValueError("a rank holds more frames than its share: the shard is not frame index mod world")
    if world == 1 and comm is None:
        table = rows
    elif comm is not None:
        table = comm.frames_allgather(rows, rows_per_rank, world)
    else:
        import torch
        pad = np.zeros((rows_per_rank, api.FRAME_ROW_FLOATS), np.float32)
        pad[:, 0] = -1
        pad[:len(rows)] = rows
        t = torch.from_numpy(pad)
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
        table = torch.cat(parts).numpy()
    out = {}
    for row in table:
        if row[0] >= 0:
            out[all_keys[int(round(float(row[0])))]] = row[1:].reshape(4, 4).copy()
    return out


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


def write_assets_dir(assets, path):
    """The Assets in the binary layout the C++ drivers read (host/Frame.h hop::Assets; host/app/run_real_all.cpp, main_realdata_auto.cpp --depth):
    model.bin, model001.bin, ppf_keys.bin, hand.txt + one cloud file per link, base_link.bin, meshes.txt + one mesh file per mesh."""
    os.makedirs(path, exist_ok=True)

    def wc(name, xyz, nrm=None):
        xyz = np.asarray(xyz, np.float32)
        nrm = np.zeros_like(xyz) if nrm is None else np.asarray(nrm, np.float32)
        with open(os.path.join(path, name), "wb") as f:
            np.array([len(xyz), 0], np.int32).tofile(f)
            np.ascontiguousarray(xyz.T).tofile(f)
            np.ascontiguousarray(nrm.T).tofile(f)

    def wm(name, V, Fi):
        with open(os.path.join(path, name), "wb") as f:
            np.array([len(V), len(Fi)], np.int32).tofile(f)
            np.ascontiguousarray(V, np.float32).tofile(f)
            np.ascontiguousarray(Fi, np.int32).tofile(f)

    wc("model.bin", *assets.model)
    wc("model001.bin", *assets.model001)
    with open(os.path.join(path, "ppf_keys.bin"), "wb") as f:
        np.array([len(assets.keys)], np.int32).tofile(f)
        np.ascontiguousarray(assets.keys, np.int32).tofile(f)
    hand = assets.hand
    with open(os.path.join(path, "hand.txt"), "w") as f:
        for name in hand.clouds:
            if name == "base_link":
                continue
            wc(f"{name}.bin", *hand.clouds[name])
            f.write(f"{name} {hand.parents[name]} {name}.bin " + " ".join(repr(float(v)) for v in hand.tf_in_parent[name].reshape(16)) + "\n")
    wc("base_link.bin", *hand.clouds["base_link"])
    with open(os.path.join(path, "meshes.txt"), "w") as f:
        wm("object.mesh", *assets.mesh)
        f.write("object object.mesh\n")
        for name, (V, Fi) in hand.meshes.items():
            wm(f"{name}.mesh", V, Fi)
            f.write(f"{name} {name}.mesh\n")
    return path


def _frame_state(ctx, cfg, assets):
    """What run_real_all.cpp builds once before its frame loop (:56-60), kept per context: the PoseEstimator (its two models and
    the lists built on them stay resident) and a second context for handbaseICP's own scene / model."""
    st = getattr(ctx, "_frame_state", None)
    if st is None or st["cfg"] is not cfg or st["assets"] is not assets:
        old_icp = st["icp_ctx"] if st else None
        st = {"cfg": cfg, "assets": assets, "est": api.PoseEstimator(cfg, assets.model, assets.model001, ctx=ctx),
              "icp_ctx": old_icp or api.Context(ctx.device)}
        ctx._frame_state = st
    return st


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
    h.setHandbaseIcpContext(_frame_state(ctx, cfg, assets)["icp_ctx"])
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
    if os.environ.get("HOP_APP_DEBUG"):
        sm = lambda a: float(np.where(np.isfinite(a), a, 1e3).astype(np.float64).sum())
        print("debug sums: scene_organized %d %.9g %.9g | scene_rgb %d %.9g %.9g | region %d %.9g | without_hand %d %.9g %.9g %.9g | mls %d %.9g %.9g | segment %d %.9g %.9g %.9g"
              % (int(fin.sum()), sm(scene_organized[fin]), sm(scene_organized_n[fin]), len(scene_rgb), sm(scene_rgb), sm(scene_rgb_n), len(rx), sm(rx), len(ox), sm(ox), sm(on), sm(oc),
                 len(mp), sm(mp), sm(mnrm), len(sx), sm(sx), sm(sn), sm(scf)))
    if os.environ.get("HOP_APP_DEBUG_DIR"):
        np.savez(os.path.join(os.environ["HOP_APP_DEBUG_DIR"], "py.npz"), ox=ox, on=on, oc=oc, mp=mp, mn=mnrm, mk=mk, sx=sx, sn=sn, scf=scf, rgb=scene_rgb[keep_n], rgbn=scene_rgb_n[keep_n])
    # :230-241
    est = _frame_state(ctx, cfg, assets)["est"]   # (run_real_all.cpp:56-60: built once, outside the frame loop)
    est.setCurScene(sx, sn, scf, cloud_withouthand_raw=cloud_withouthand_raw, depth_raw=depth_raw, depth_unit=depth_unit, K=K)
    est.registerHandMesh(h)
    est.registerMesh(assets.mesh[0], assets.mesh[1], "object")
    if not est.runSuper4pcs(assets.keys):
        return ident
    info["n_generated"] = ctx.hypos_count()
    est.clusterPoses(30, 0.015, True)
    info["n_clusters"] = ctx.hypos_count()
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
        class _Dumper(yaml.SafeDumper):     # lists in flow style ("cam_K: [..]"), mappings in block style: the form of the reference's file
            pass
        _Dumper.add_representer(list, lambda d, data: d.represent_sequence("tag:yaml.org,2002:seq", data, flow_style=True))
        yaml.dump(cfg, f, Dumper=_Dumper, default_flow_style=False)
    fmt = lambda T: "\n".join(" ".join(repr(float(v)) for v in row) for row in np.asarray(T, np.float64)) + "\n"
    for k in range(n_frames):
        g = synth.grasp_depth_frame(seed=seed0 + k)
        Image.fromarray(g["depth"]).save(os.path.join(rec, f"depth{k}.png"))
        Image.fromarray(np.zeros((g["depth"].shape[0], g["depth"].shape[1], 3), np.uint8)).save(os.path.join(rec, f"rgb{k}.png"))
        open(os.path.join(rec, f"palm_in_base{k}.txt"), "w").write(fmt(g["handbase_in_cam_reported"]))
        open(os.path.join(rec, f"arm_left_link_7_t_{k}.txt"), "w").write(fmt(np.eye(4)))
        open(os.path.join(rec, "refined_gt", f"ob_in_cam{k}.txt"), "w").write(fmt(g["object_in_cam"]))
    return rec


def run_raw(base_dir, cfg, model_name=None, records=None, rank=0, world=1, device=0, ctx=None, assets=None, force=False, use_physics=True, use_render=True,
            poses_out=None):
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
                if poses_out is not None:
                    poses_out[(record, idx)] = np.loadtxt(out).astype(np.float32)
                continue                                            # resume: the frame was finished by an earlier run
            leftarm_in_base = parse_pose_txt(os.path.join(rec, f"arm_left_link_7_t_{idx}.txt"))
            palm_in_baselink = parse_pose_txt(os.path.join(rec, f"palm_in_base{idx}.txt"))
            depth = read_depth_png(os.path.join(rec, f"depth{idx}.png"))
            pose = process_frame(ctx, cfg, assets, depth, K, handbase_in_cam_of(cfg, leftarm_in_base, palm_in_baselink), 0.001, use_physics, use_render)
            os.makedirs(os.path.dirname(out), exist_ok=True)
            np.savetxt(out + ".tmp", pose.astype(np.float64))
            os.replace(out + ".tmp", out)                           # a killed run never leaves a half-written result
            done[record].append(idx)
            if poses_out is not None:
                poses_out[(record, idx)] = pose
    if own:
        ctx.close()
    return done


def eval_raw(base_dir, model_name, model_pts, pred=None):
    """scripts/eval_all.py:36-79 over every record directory of one object; `pred` ({(record, idx): pose}, the gathered table) replaces
    reading predict/<idx>/model2scene.txt back."""
    errs = {}
    mdir = os.path.join(base_dir, model_name)
    for record in sorted(d for d in os.listdir(mdir) if os.path.isdir(os.path.join(mdir, d))):
        rec = os.path.join(mdir, record)
        for idx in raw_frame_indices(rec):
            gt_file = os.path.join(rec, "refined_gt", f"ob_in_cam{idx}.txt")
            pred_file = os.path.join(rec, "predict", str(idx), "model2scene.txt")
            if pred is not None:
                pr = np.asarray(pred.get((record, idx), np.eye(4)), np.float64)
            else:
                pr = np.loadtxt(pred_file) if os.path.exists(pred_file) else np.eye(4)
            gt = parse_pose_txt(gt_file).astype(np.float64) if os.path.exists(gt_file) else np.eye(4)
            errs[(record, idx)] = adi(pr[:3, :3], pr[:3, 3], gt[:3, :3], gt[:3, 3], np.asarray(model_pts, np.float64))
    e = np.array(list(errs.values()))
    n = max(len(e), 1)
    return {"total": int(len(e)), "recall_5mm": float(np.sum(e < 0.005) / n), "recall_10mm": float(np.sum(e < 0.010) / n), "errs": errs}


def adi(R_est, t_est, R_gt, t_gt, pts):
    """scripts/eval_utils.py:181-200."""
    pts_est = pts @ R_est.T + t_est
    pts_gt = pts @ R_gt.T + t_gt
    nn_dists, _ = cKDTree(pts_est).query(pts_gt, k=1)
    return float(nn_dists.mean())


def eval_all(record_dir, model_pts, pred=None):
    """scripts/eval_all.py:36-79 for one object: ADI of every frame that has a ground truth (a missing prediction counts
    as the identity, as there), recall at 5 mm (the authors' threshold) and at 10 mm."""
    errs = {}
    for idx in frame_indices(record_dir):
        gt_file = os.path.join(record_dir, "refined_gt", f"ob_in_cam{idx}.txt")
        pred_file = os.path.join(record_dir, "predict", str(idx), "model2scene.txt")
        if pred is not None:
            pr = np.asarray(pred.get(idx, np.eye(4)), np.float64)
        else:
            pr = np.loadtxt(pred_file) if os.path.exists(pred_file) else np.eye(4)
        gt = np.loadtxt(gt_file) if os.path.exists(gt_file) else np.eye(4)
        errs[idx] = adi(pr[:3, :3], pr[:3, 3], gt[:3, :3], gt[:3, 3], np.asarray(model_pts, np.float64))
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
    dist, comm = None, None
    if world > 1:
        import torch
        import torch.distributed as dist
        local = local % max(torch.cuda.device_count(), 1)  # self-test: several ranks on one GPU (HOP_BENCH_BACKEND=gloo)
        torch.cuda.set_device(local)
        backend = os.environ.get("HOP_BENCH_BACKEND", "nccl")
        dist.init_process_group(backend=backend)
        if backend == "nccl":
            # the library's own RCCL communicator for the final gather: rank 0's unique id travels through torch.distributed
            idt = torch.zeros(128, dtype=torch.uint8, device=torch.device("cuda", local))
            if rank == 0:
                idt = torch.tensor(list(api.Comm.unique_id()), dtype=torch.uint8, device=torch.device("cuda", local))
            dist.broadcast(idt, 0)
            comm = api.Comm(local, bytes(idt.cpu().numpy().tolist()), rank, world)
    if args.base:
        from . import config as hop_config
        if args.synthetic and rank == 0:
            write_synthetic_record(args.base, args.model or "ellipse", n_frames=args.synthetic)
        if dist is not None:
            dist.barrier()
        cfg_path = args.config or (os.path.join(args.base, "config_autodataset.yaml") if os.path.exists(os.path.join(args.base, "config_autodataset.yaml")) else None)
        cfg = hop_config.load_config(cfg_path)
        model_name = args.model or cfg["model_name"]
        if model_name != "ellipse":
            #  only the synthetic ellipse has assets here: the paper's meshes / PPF tables are download links
            raise SystemExit(f"no assets for model '{model_name}': this runner ships the synthetic ellipse only (pass --model ellipse or supply Assets through run_raw)")
        local_poses = {}
        done = run_raw(args.base, cfg, model_name, rank=rank, world=world, device=local, force=args.force, poses_out=local_poses)
        mdir = os.path.join(args.base, model_name)
        all_keys = [(rec, i) for rec in sorted(d for d in os.listdir(mdir) if os.path.isdir(os.path.join(mdir, d))) for i in raw_frame_indices(os.path.join(mdir, rec))]
        gathered = gather_frame_poses(local_poses, all_keys, rank, world, comm=comm, dist=dist)   # the one collective of the run
        if rank == 0:
            r = eval_raw(args.base, model_name, Assets().model001[0], pred=gathered)
            r.pop("errs")
            print(json.dumps({"frames_this_rank": sum(len(v) for v in done.values()), "frames_gathered": len(gathered), "world": world,
                              "gather": "hop_frames_allgather (RCCL)" if comm is not None else ("torch.distributed all_gather" if dist is not None else "none"), **r}))
    else:
        if args.synthetic and rank == 0:
            write_synthetic_dataset(args.root, args.synthetic)
        if dist is not None:
            dist.barrier()
        local_poses = {}
        done = run(args.root, rank=rank, world=world, device=local, poses_out=local_poses)
        gathered = gather_frame_poses(local_poses, frame_indices(args.root), rank, world, comm=comm, dist=dist)
        if rank == 0:
            r = eval_all(args.root, synth.ellipsoid_model(4000)[0], pred=gathered)
            r.pop("errs")
            print(json.dumps({"frames_this_rank": len(done), "frames_gathered": len(gathered), "world": world,
                              "gather": "hop_frames_allgather (RCCL)" if comm is not None else ("torch.distributed all_gather" if dist is not None else "none"), **r}))
    if comm is not None:
        comm.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
