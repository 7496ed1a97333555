"""Dataset runner and evaluator: the frame loop of the reference's
src/perception/src/app/run_real_all.cpp:70-273 above the C-ABI, sharded over GPUs frame by frame, and the authors'
evaluator scripts/eval_all.py:11-79 with scripts/eval_utils.py:181-200 (ADI) next to it.

Directory layout (the reference's, with the prepared clouds of a frame in one file instead of rgb / depth images plus
the preprocessing of run_real_all.cpp:100-190):

    <record_dir>/cloud<idx>.npz                  xyz, nrm, conf of `object_segment` (what est.setCurScene receives)
    <record_dir>/refined_gt/ob_in_cam<idx>.txt   ground-truth pose, 4x4
    <record_dir>/predict/<idx>/model2scene.txt   written here, read by the evaluator

    python tools/run_real_all.py --root DIR [--synthetic N]                                   (1 GPU)
    python -m torch.distributed.run --nproc-per-node N tools/run_real_all.py --root DIR       (one rank per GPU)

Frame idx goes to rank idx mod world; nothing is exchanged between ranks, rank 0 evaluates after a
barrier.  The datasets of the paper are not redistributed with the reference, so `write_synthetic_dataset` emits frames
of the synthetic ellipse in this layout."""
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
    ctx.icp_refine(10, 45.0, 0.01, max_hypotheses=100, nn_mode=3)
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
    ap.add_argument("--root", required=True, help="record directory (cloud<idx>.npz, refined_gt/)")
    ap.add_argument("--synthetic", type=int, default=0, help="write this many synthetic frames into --root first (rank 0)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        local = local % max(torch.cuda.device_count(), 1)  # self-test: several ranks on one GPU (HOP_BENCH_BACKEND=gloo)
        torch.cuda.set_device(local)
        dist.init_process_group(backend=os.environ.get("HOP_BENCH_BACKEND", "nccl"))
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
