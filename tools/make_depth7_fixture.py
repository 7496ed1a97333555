#!/usr/bin/env python
"""tools/make_depth7_fixture.py -- C1 fixture (BASELINE.json configs[0], SURVEY.md 8(d) "C1-synthetic-model").

Run in the BUILD container only (it reads the reference's example data); emits tests/golden/depth7_hand_region.npz,
a small data fixture: the hand-region cloud of example/depth7.png in the camera frame.  The steps restate the scene
preparation of main_realdata_auto.cpp:44-96 with plain numpy (the reference uses PCL/OpenCV, absent here):

  depth (uint16 mm) -> metres, keep 0.1..2.0 m                  (Utils.cpp:36-55, SR300_DEPTH_UNIT)
  back-projection with cam_K                                      (Utils::convert3dOrganizedRGB)
  handbase_in_cam = cam1_in_leftarm^-1 * leftarm_in_base^-1 * palm_in_baselink * handbase_in_palm   (main :44-49)
  1 mm voxel thinning, crop in the hand-base frame z in [-0.12, 0.05], x in [-0.25, -0.07], y in [-0.2, 0.2]  (main :66-92)
  3 mm voxel thinning (what Hand::setCurScene works on), normals from a PCA of the 12 nearest neighbours, oriented
  towards the camera (the reference uses an integral-image estimator on the organised cloud: the fixture only needs
  plausible unit normals; both the GPU path and the oracle receive the same arrays).

The object and hand models are the synthetic stand-ins (the meshes are not part of the reference repository)."""
import os
import sys

import numpy as np
import yaml
from PIL import Image
from scipy.spatial import cKDTree

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def quat_to_R(x, y, z, w):
    n = np.sqrt(x * x + y * y + z * z + w * w)
    x, y, z, w = x / n, y / n, z / n, w / n
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def voxel_first(xyz, size):
    key = np.floor(xyz / size).astype(np.int64)
    _, idx = np.unique(key, axis=0, return_index=True)
    return np.sort(idx)


def main():
    cfg = yaml.safe_load(open(os.path.join(REF, "config_autodataset.yaml")))
    K = np.array(cfg["cam_K"], np.float64).reshape(3, 3)
    d = cfg["cam1_in_leftarm"]
    cam1_in_leftarm = np.eye(4)
    cam1_in_leftarm[:3, 3] = d[:3]
    cam1_in_leftarm[:3, :3] = quat_to_R(d[3], d[4], d[5], d[6])
    handbase_in_palm = np.array(cfg["handbase_in_palm"], np.float64).reshape(4, 4)
    handbase_in_palm[3] = [0, 0, 0, 1]
    palm_in_baselink = np.loadtxt(os.path.join(REF, "example", "palm_in_base7.txt")).reshape(4, 4)
    leftarm_in_base = np.loadtxt(os.path.join(REF, "example", "arm_left_link_7_t_7.txt")).reshape(4, 4)
    handbase_in_cam = np.linalg.inv(cam1_in_leftarm) @ np.linalg.inv(leftarm_in_base) @ palm_in_baselink @ handbase_in_palm

    depth = np.array(Image.open(os.path.join(REF, "example", "depth7.png"))).astype(np.float64) * 1e-3
    depth[(depth > 2.0) | (depth < 0.1)] = 0.0
    v, u = np.nonzero(depth)
    z = depth[v, u]
    cam = np.stack([(u - K[0, 2]) * z / K[0, 0], (v - K[1, 2]) * z / K[1, 1], z], axis=1)
    n_valid = len(cam)
    cam = cam[voxel_first(cam, 0.001)]
    n_1mm = len(cam)
    Tinv = np.linalg.inv(handbase_in_cam)
    hb = cam @ Tinv[:3, :3].T + Tinv[:3, 3]
    keep = (hb[:, 2] >= -0.12) & (hb[:, 2] <= 0.05) & (hb[:, 0] >= -0.25) & (hb[:, 0] <= -0.07) & (hb[:, 1] >= -0.2) & (hb[:, 1] <= 0.2)
    cam = cam[keep]
    n_crop = len(cam)
    cam = cam[voxel_first(cam, 0.003)]
    tree = cKDTree(cam)
    _, nb = tree.query(cam, k=min(12, len(cam)))
    nrm = np.zeros_like(cam)
    for i in range(len(cam)):
        q = cam[nb[i]] - cam[nb[i]].mean(axis=0)
        w, vec = np.linalg.eigh(q.T @ q)
        n = vec[:, 0]
        nrm[i] = -n if np.dot(n, cam[i]) > 0 else n   # towards the camera at the origin
    out = os.path.join(ROOT, "tests", "golden", "depth7_hand_region.npz")
    np.savez_compressed(out, xyz=cam.astype(np.float32), nrm=nrm.astype(np.float32), handbase_in_cam=handbase_in_cam.astype(np.float32),
                        counts=np.array([n_valid, n_1mm, n_crop, len(cam)], np.int64))
    # the raw frame itself (data of the reference's example/ directory) for the scene front end (hop_scene_from_depth)
    raw = np.array(Image.open(os.path.join(REF, "example", "depth7.png"))).astype(np.uint16)
    out_raw = os.path.join(ROOT, "tests", "golden", "depth7_raw.npz")
    np.savez_compressed(out_raw, depth=raw, K=K.astype(np.float32), handbase_in_cam=handbase_in_cam.astype(np.float32),
                        cam_in_handbase=np.linalg.inv(handbase_in_cam).astype(np.float32))
    print("raw depth written", out_raw, os.path.getsize(out_raw), "bytes")
    print("valid px", n_valid, "-> 1 mm", n_1mm, "-> crop", n_crop, "-> 3 mm", len(cam), "written", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    sys.exit(main())
