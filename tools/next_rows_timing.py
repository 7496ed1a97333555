#!/usr/bin/env python
"""Wall time (host call to host result, transfers included) of the rows N2, N3a-N3g, and of a whole frame from a depth image;
prints one JSON object.  python tools/next_rows_timing.py"""
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hop_loader  # noqa: E402

hop = hop_loader.load()
from hop_amd import api, config as hop_config  # noqa: E402

synth = hop.synth


def timed(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(1e3 * (time.perf_counter() - t0))
    return float(np.median(t))


def main():
    ctx = api.Context(0)
    out = {}
    g = np.load(os.path.join(ROOT, "tests", "golden", "depth7_raw.npz"))
    out["N3b scene_from_depth (640x480 example frame)"] = timed(lambda: ctx.scene_from_depth(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"]))
    sc = synth.make_scene(30000, seed=3)
    out["N3c object_segment (30 k dense points)"] = timed(lambda: ctx.object_segment(sc.xyz, sc.nrm, sc.conf, 0.003))
    hand = synth.t42_hand()
    ang = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}
    hxyz, hnrm = synth.make_hand_scene(hand, ang, 20000, seed=5)
    I4 = np.eye(4, dtype=np.float32)
    out["N3d hand_scene_filters (20 k points)"] = timed(lambda: ctx.hand_scene_filters(hxyz, hnrm, I4))
    out["N3e voxel_downsample_normals (20 k points, 5 mm)"] = timed(lambda: ctx.voxel_downsample_normals(hxyz, hnrm, 0.005))
    cfg = hop_config.load_config(os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml"))
    h = api.HandT42(cfg, hand, ctx=ctx)
    for n, a in ang.items():
        h._tf_self[n] = synth.rx(a).astype(np.float32)
    h.makeHandCloud()
    scn = synth.make_scene(20000, seed=9)
    T = synth.se3(synth.rot_from_axis_angle([1.0, 0.2, -0.1], 2.6), [0.03, -0.02, 0.55]).astype(np.float32)
    out["N3a remove_surrounding (20 k points, 5 links)"] = timed(lambda: h.removeSurroundingPointsAndAssignProbability(scn.xyz, scn.nrm, T, 0.003 ** 2))
    out["N3g scene_from_depth_normals (640x480 example frame, integral-image normals through grid and crop)"] = timed(
        lambda: ctx.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"]))
    org = api.organized_cloud(g["depth"], g["K"])
    out["N3g normals_integral_image (640x480 organised cloud)"] = timed(lambda: ctx.normals_integral_image(org))
    mx, _ = synth.ellipsoid_model(12000)
    out["N3g normals_mls (12 k points, 3 mm)"] = timed(lambda: ctx.normals_mls(mx, 0.003, 2))
    dense = synth.make_scene(40000, seed=4).xyz
    out["N3g normals_mls (40 k points, 3 mm)"] = timed(lambda: ctx.normals_mls(dense, 0.003, 2))
    gf = synth.grasp_depth_frame(seed=2)
    hV, hF = gf["hand_mesh_cam"]
    poses = synth.replay_poses(gf["object_in_cam"], 64, seed=5, max_rot_deg=20.0, max_trans=0.02)
    ctx.render_set_object(gf["object_V"], gf["object_F"])
    out["N2 render_set_frame (640x480 depth, 316-face hand)"] = timed(lambda: ctx.render_set_frame(gf["depth"], 0.001, gf["K"], hV, hF))

    def rej(mode):
        ctx.hypos_upload(poses)
        ctx.reject_by_render(2.0, 0.3, sum_mode=mode)
    out["N2 reject_by_render (64 hypotheses, 1280-face object), ordered sums"] = timed(lambda: rej(0))
    out["N2 reject_by_render (64 hypotheses, 1280-face object), reduced sums"] = timed(lambda: rej(1))
    from hop_amd import run_real_all as rr
    assets = rr.Assets()
    out["whole frame from a depth image (run_real_all.process_frame, as-shipped chain, both rejections)"] = timed(
        lambda: rr.process_frame(ctx, cfg, assets, gf["depth"], gf["K"], gf["handbase_in_cam_reported"]), reps=3)
    print(json.dumps({"unit": "ms per call, median of 5, host to host", **out}))


if __name__ == "__main__":
    main()
