"""Seeded synthetic inputs for the hot path.

The reference ships neither meshes, nor the PPF table, nor the hand URDF
(README.md:49-54 of the reference are download links), so every configuration that
needs them is run on the stand-ins built here:

* ``ellipsoid_model``  -- the "ellipse" object: semi-axes (0.040, 0.025, 0.020) m, Fibonacci
  points on the surface, analytic outward normals.
* ``make_scene``       -- the model under a random SE(3), camera-facing half only, Gaussian noise,
  hand-like clutter, per-point confidence.
* ``ppf_key_table``    -- the PPF key set, following the recipe of the reference's
  ``src/perception/src/app/computePPF.cpp:56-107`` (keys of ordered pairs i<j of the 5 mm model).
* ``t42_hand``         -- a parametric stand-in for the Yale T42 hand: 4 finger links (boxes sampled
  at 5 mm), each rotating about its local x axis, with the reference's link names.
* ``ellipsoid_mesh`` / ``box_mesh`` / ``torus_mesh`` / ``lshape_mesh`` -- closed, outward-oriented triangle
  meshes (row N1: the object mesh and the finger links' convex meshes the reference loads from OBJ files).
* ``physics_case``     -- row N1: a grasp of the ellipsoid by the stand-in hand with every input of
  ``rejectByCollisionOrNonTouching`` and hypotheses that exercise each of its checks.
* ``replay_poses``     -- ground truth composed with bounded perturbations (rot <= 30 deg,
  trans <= 15 mm) so that scoring kernels can be timed independently of the generator.

Everything is numpy + ``np.random.Generator(PCG64(seed))``; all outputs are float32 C-contiguous.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

SEMI_AXES = (0.040, 0.025, 0.020)


# --------------------------------------------------------------------------- geometry helpers
def rot_from_axis_angle(axis, angle):
    axis = np.asarray(axis, dtype=np.float64)
    axis = axis / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + math.sin(angle) * K + (1 - math.cos(angle)) * (K @ K)


def random_rotation(rng):
    """Uniform rotation from a unit quaternion."""
    q = rng.normal(size=4)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def se3(R, t):
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = t
    return T


def apply(T, xyz):
    return (xyz.astype(np.float64) @ T[:3, :3].T + T[:3, 3]).astype(np.float32)


def rotate(T, nrm):
    return (nrm.astype(np.float64) @ T[:3, :3].T).astype(np.float32)


def voxel_thin(xyz, leaf, *others):
    """Keep the first point of every occupied voxel (input order preserved)."""
    key = np.floor(xyz.astype(np.float64) / leaf).astype(np.int64)
    key -= key.min(axis=0)
    dims = key.max(axis=0) + 1
    lin = (key[:, 0] * dims[1] + key[:, 1]) * dims[2] + key[:, 2]
    _, first = np.unique(lin, return_index=True)
    first.sort()
    return (xyz[first],) + tuple(o[first] for o in others)


# --------------------------------------------------------------------------- object model
def ellipsoid_model(m: int, semi=SEMI_AXES):
    """``m`` Fibonacci-sphere points mapped onto the ellipsoid; returns (xyz, normals) float32."""
    a, b, c = semi
    i = np.arange(m, dtype=np.float64) + 0.5
    phi = np.arccos(1 - 2 * i / m)
    theta = math.pi * (1 + 5**0.5) * i
    u = np.stack([np.cos(theta) * np.sin(phi), np.sin(theta) * np.sin(phi), np.cos(phi)], axis=1)
    xyz = u * np.array([a, b, c])
    n = xyz / np.array([a * a, b * b, c * c])
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    return xyz.astype(np.float32), n.astype(np.float32)


def ellipsoid_model_spacing(spacing: float, semi=SEMI_AXES):
    """Model with approximately ``spacing`` metres between neighbours (5 mm / 1 mm levels)."""
    a, b, c = semi
    p = 1.6075
    area = 4 * math.pi * (((a * b) ** p + (a * c) ** p + (b * c) ** p) / 3) ** (1 / p)
    m = max(16, int(round(area / (spacing * spacing))))
    return ellipsoid_model(m, semi)


# --------------------------------------------------------------------------- scene
@dataclass
class Scene:
    xyz: np.ndarray  # (N,3) float32, camera frame
    nrm: np.ndarray  # (N,3) float32, unit, towards the camera
    conf: np.ndarray  # (N,)  float32
    gt_pose: np.ndarray  # (4,4) float64 model->camera
    n_object: int = 0


def make_scene(n_points: int, seed: int = 7, semi=SEMI_AXES, noise=0.0005, clutter_frac=0.12,
               normal_jitter_deg=3.0, high_conf_only=True) -> Scene:
    """Camera-facing half of the ellipsoid under a random SE(3) + clutter.

    t in [-0.1,0.1]^2 x [0.3,0.5], uniform rotation.  ``n_points`` is the size of the returned cloud;
    about ``clutter_frac`` of it are hand-like clutter points next to the object.  With
    ``high_conf_only`` every returned point has confidence >= 0.8 (i.e. the cloud is what the reference
    calls ``_scene_high_confidence``, PoseEstimator.cpp:41-45); otherwise clutter gets 0.3.
    """
    rng = np.random.Generator(np.random.PCG64(seed))
    R = random_rotation(rng)
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.3, 0.5)])
    T = se3(R, t)
    n_clutter = int(round(n_points * clutter_frac))
    n_obj = n_points - n_clutter
    # over-sample the full surface, keep the visible half, then cut to n_obj
    dense_xyz, dense_n = ellipsoid_model(int(n_obj * 2.6) + 64, semi)
    perm = rng.permutation(len(dense_xyz))
    dense_xyz, dense_n = dense_xyz[perm], dense_n[perm]
    xyz = apply(T, dense_xyz).astype(np.float64)
    nrm = rotate(T, dense_n).astype(np.float64)
    view = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
    vis = np.einsum("ij,ij->i", nrm, view) < -0.05
    xyz, nrm = xyz[vis][:n_obj], nrm[vis][:n_obj]
    if len(xyz) < n_obj:
        raise RuntimeError("not enough visible points; raise the over-sampling factor")
    xyz = xyz + rng.normal(scale=noise, size=xyz.shape)
    # normal jitter
    jit = rng.normal(scale=math.radians(normal_jitter_deg), size=nrm.shape)
    nrm = nrm + np.cross(jit, nrm)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    # clutter: two finger-like slabs touching the object along +-x of the object frame
    cl = []
    cn = []
    for s in (-1.0, 1.0):
        k = n_clutter // 2 if s < 0 else n_clutter - n_clutter // 2
        local = np.stack(
            [np.full(k, s * (semi[0] + 0.004)) + rng.normal(scale=0.0004, size=k),
             rng.uniform(-0.012, 0.012, size=k), rng.uniform(-0.03, 0.03, size=k)], axis=1)
        # what the camera sees of a finger is its top, not the contact face: normals tangent to the object there
        ln = np.tile(np.array([0.0, 0.0, 1.0]), (k, 1))
        cl.append(local)
        cn.append(ln)
    cl = apply(T, np.concatenate(cl)).astype(np.float64)
    cn = rotate(T, np.concatenate(cn)).astype(np.float64)
    # orient clutter normals towards the camera like flipNormalTowardsViewpoint does
    flip = np.einsum("ij,ij->i", cn, cl) > 0
    cn[flip] *= -1
    all_xyz = np.concatenate([xyz, cl]).astype(np.float32)
    all_nrm = np.concatenate([nrm, cn]).astype(np.float32)
    conf = np.concatenate([np.ones(len(xyz)), np.full(len(cl), 0.85 if high_conf_only else 0.3)]).astype(np.float32)
    # interleave deterministically so clutter is not a suffix
    order = rng.permutation(len(all_xyz))
    return Scene(np.ascontiguousarray(all_xyz[order]), np.ascontiguousarray(all_nrm[order]),
                 np.ascontiguousarray(conf[order]), T, n_obj)


# --------------------------------------------------------------------------- the other objects of config_autodataset.yaml:35-59
# Stand-ins for BASELINE configs[2] ("cuboid + cylinder + tless"): the reference's meshes are download links (README.md:49-54).  Each
# has the symmetry class its `object_symmetry` entry declares, so that clusterPoses' per-axis folding (PoseEstimator.cpp:135-196)
# is exercised with 90 (cuboid z), 0 (cylinder z, tless3 y) and 360 (no symmetry) next to the ellipse's 180.
OBJECT_SYMMETRY = {"ellipse": (180.0, 180.0, 180.0), "cuboid": (180.0, 180.0, 90.0), "cylinder": (180.0, 180.0, 0.0),
                   "tless3": (360.0, 0.0, 360.0), "mustard": (360.0, 360.0, 360.0)}


def _plane_patch(origin, eu, ev, nu, nv, normal):
    u = (np.arange(nu) + 0.5) / nu
    v = (np.arange(nv) + 0.5) / nv
    uu, vv = np.meshgrid(u, v, indexing="ij")
    P = np.asarray(origin)[None] + uu.reshape(-1, 1) * np.asarray(eu)[None] + vv.reshape(-1, 1) * np.asarray(ev)[None]
    return P, np.tile(np.asarray(normal, np.float64), (len(P), 1))


def _box_surface(lo, hi, spacing):
    lo, hi = np.asarray(lo, np.float64), np.asarray(hi, np.float64)
    P, N = [], []
    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        ext = hi - lo
        nu, nv = max(1, int(round(ext[o[0]] / spacing))), max(1, int(round(ext[o[1]] / spacing)))
        for val, sgn in ((lo[ax], -1.0), (hi[ax], 1.0)):
            org = lo.copy()
            org[ax] = val
            eu, ev, nn = np.zeros(3), np.zeros(3), np.zeros(3)
            eu[o[0]], ev[o[1]], nn[ax] = ext[o[0]], ext[o[1]], sgn
            a, b = _plane_patch(org, eu, ev, nu, nv, nn)
            P.append(a), N.append(b)
    return np.concatenate(P), np.concatenate(N)


def _cylinder_surface(radius, z0, z1, spacing, axis=2, caps=(True, True), inner_cap_radius=(0.0, 0.0)):
    """side + caps of a cylinder about `axis`; a cap can be an annulus (inner radius > 0)"""
    P, N = [], []
    nz = max(1, int(round((z1 - z0) / spacing)))
    nt = max(8, int(round(2 * math.pi * radius / spacing)))
    zz = z0 + (np.arange(nz) + 0.5) / nz * (z1 - z0)
    tt = (np.arange(nt) + 0.5) / nt * 2 * math.pi
    T, Z = np.meshgrid(tt, zz, indexing="ij")
    side = np.stack([radius * np.cos(T).ravel(), radius * np.sin(T).ravel(), Z.ravel()], axis=1)
    sn = np.stack([np.cos(T).ravel(), np.sin(T).ravel(), np.zeros(T.size)], axis=1)
    P.append(side), N.append(sn)
    for k, (z, sgn) in enumerate(((z0, -1.0), (z1, 1.0))):
        if not caps[k]:
            continue
        r_in = inner_cap_radius[k]
        nr = max(1, int(round((radius - r_in) / spacing)))
        for i in range(nr):
            r = r_in + (i + 0.5) / nr * (radius - r_in)
            m = max(6, int(round(2 * math.pi * r / spacing)))
            t = (np.arange(m) + 0.5 * (i % 2)) / m * 2 * math.pi
            P.append(np.stack([r * np.cos(t), r * np.sin(t), np.full(m, z)], axis=1))
            N.append(np.tile(np.array([0.0, 0.0, sgn]), (m, 1)))
    P, N = np.concatenate(P), np.concatenate(N)
    if axis != 2:  # z -> the requested axis (a cyclic permutation keeps the orientation)
        perm = {0: [2, 0, 1], 1: [1, 2, 0]}[axis]
        P, N = P[:, perm], N[:, perm]
    return P, N


def object_surface(name: str, spacing: float):
    """(xyz, outward unit normals) float32 of the stand-in object `name` in its model frame, ~`spacing` between neighbours."""
    if name == "ellipse":
        return ellipsoid_model_spacing(spacing)
    if name == "cuboid":       # square cross-section in x / y (90 degrees about z), 180 about x and y
        P, N = _box_surface((-0.025, -0.025, -0.04), (0.025, 0.025, 0.04), spacing)
    elif name == "cylinder":   # any angle about z, 180 about x and y
        P, N = _cylinder_surface(0.022, -0.045, 0.045, spacing)
    elif name == "tless3":     # two coaxial cylinders about y: any angle about y, nothing else
        a, an = _cylinder_surface(0.030, -0.030, 0.000, spacing, axis=1, caps=(True, True), inner_cap_radius=(0.0, 0.016))
        b, bn = _cylinder_surface(0.016, 0.000, 0.035, spacing, axis=1, caps=(False, True))
        P, N = np.concatenate([a, b]), np.concatenate([an, bn])
    elif name == "mustard":    # no symmetry: a box with an off-centre box on top and a side handle
        parts = [_box_surface((-0.03, -0.02, -0.045), (0.03, 0.02, 0.03), spacing), _box_surface((-0.03, -0.012, 0.03), (-0.005, 0.012, 0.06), spacing),
                 _box_surface((0.03, -0.008, -0.02), (0.042, 0.008, 0.005), spacing)]
        P, N = np.concatenate([q[0] for q in parts]), np.concatenate([q[1] for q in parts])
        # drop the faces of a part that lie inside another part (they are not on the surface of the union)
        def inside(pts, lo, hi, eps=1e-6):
            return np.all((pts > np.asarray(lo) + eps) & (pts < np.asarray(hi) - eps), axis=1)
        boxes = [((-0.03, -0.02, -0.045), (0.03, 0.02, 0.03)), ((-0.03, -0.012, 0.03), (-0.005, 0.012, 0.06)), ((0.03, -0.008, -0.02), (0.042, 0.008, 0.005))]
        keep = np.ones(len(P), bool)
        for lo, hi in boxes:
            keep &= ~inside(P, lo, hi)
        # faces glued to a neighbour: the contact patches z = 0.03 (top box on the body) and x = 0.03 (handle on the body)
        glue = (np.isclose(P[:, 2], 0.03) & (P[:, 0] > -0.03) & (P[:, 0] < -0.005) & (np.abs(P[:, 1]) < 0.012)) | \
               (np.isclose(P[:, 0], 0.03) & (np.abs(P[:, 1]) < 0.008) & (P[:, 2] > -0.02) & (P[:, 2] < 0.005))
        P, N = P[keep & ~glue], N[keep & ~glue]
    else:
        raise ValueError(name)
    return np.ascontiguousarray(P, np.float32), np.ascontiguousarray(N, np.float32)


def object_model(name: str, spacing: float):
    """Model cloud of the stand-in at a voxel level (5 mm / 1 mm): surface samples thinned to one per voxel."""
    xyz, nrm = object_surface(name, spacing * 0.5)
    xyz, nrm = voxel_thin(xyz, spacing, nrm)
    return np.ascontiguousarray(xyz), np.ascontiguousarray(nrm)


def make_object_scene(name: str, n_points: int, seed: int = 7, noise=0.0004, clutter_frac=0.1, normal_jitter_deg=3.0) -> Scene:
    """make_scene for any stand-in object: the camera-facing part of its surface (normal test; self-occlusion of the two non-convex
    stand-ins is ignored) under a random SE(3), noise, finger-like clutter slabs next to it."""
    if name == "ellipse":
        return make_scene(n_points, seed=seed)
    rng = np.random.Generator(np.random.PCG64(seed))
    R = random_rotation(rng)
    t = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), rng.uniform(0.3, 0.5)])
    T = se3(R, t)
    n_clutter = int(round(n_points * clutter_frac))
    n_obj = n_points - n_clutter
    sp = 0.0012
    while True:
        dx, dn = object_surface(name, sp)
        perm = rng.permutation(len(dx))
        xyz = apply(T, dx[perm]).astype(np.float64)
        nrm = rotate(T, dn[perm]).astype(np.float64)
        view = xyz / np.linalg.norm(xyz, axis=1, keepdims=True)
        vis = np.einsum("ij,ij->i", nrm, view) < -0.05
        if vis.sum() >= n_obj:
            break
        sp *= 0.8
    xyz, nrm = xyz[vis][:n_obj], nrm[vis][:n_obj]
    xyz = xyz + rng.normal(scale=noise, size=xyz.shape)
    jit = rng.normal(scale=math.radians(normal_jitter_deg), size=nrm.shape)
    nrm = nrm + np.cross(jit, nrm)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    ext = np.abs(dx).max(axis=0)
    cl, cn = [], []
    for s_ in (-1.0, 1.0):
        k = n_clutter // 2 if s_ < 0 else n_clutter - n_clutter // 2
        local = np.stack([np.full(k, s_ * (ext[0] + 0.004)) + rng.normal(scale=0.0004, size=k), rng.uniform(-0.012, 0.012, size=k),
                          rng.uniform(-0.03, 0.03, size=k)], axis=1)
        cl.append(local)
        cn.append(np.tile(np.array([0.0, 0.0, 1.0]), (k, 1)))
    cl = apply(T, np.concatenate(cl)).astype(np.float64)
    cn = rotate(T, np.concatenate(cn)).astype(np.float64)
    flip = np.einsum("ij,ij->i", cn, cl) > 0
    cn[flip] *= -1
    all_xyz = np.concatenate([xyz, cl]).astype(np.float32)
    all_nrm = np.concatenate([nrm, cn]).astype(np.float32)
    conf = np.concatenate([np.ones(len(xyz)), np.full(len(cl), 0.85)]).astype(np.float32)
    order = rng.permutation(len(all_xyz))
    return Scene(np.ascontiguousarray(all_xyz[order]), np.ascontiguousarray(all_nrm[order]), np.ascontiguousarray(conf[order]), T, n_obj)


def symmetry_rotations(name: str, steps_for_continuous: int = 72):
    """Rotations of the model frame that map the stand-in onto itself (for pose comparison modulo symmetry): products of the per-axis
    groups of OBJECT_SYMMETRY (0 = continuous, sampled)."""
    groups = []
    for ax, deg in enumerate(OBJECT_SYMMETRY[name]):
        if deg >= 360.0:
            angles = [0.0]
        elif deg == 0.0:
            angles = [2 * math.pi * k / steps_for_continuous for k in range(steps_for_continuous)]
        else:
            angles = [math.radians(deg) * k for k in range(int(round(360.0 / deg)))]
        e = np.zeros(3)
        e[ax] = 1.0
        groups.append([rot_from_axis_angle(e, a) for a in angles])
    out = []
    for a in groups[0]:
        for b in groups[1]:
            for c in groups[2]:
                out.append(a @ b @ c)
    return out


# --------------------------------------------------------------------------- PPF key table
_DIST_DISCRET = 5
_ANGLE_DISCRET = 10


def _closest_bin(value, disc):
    lower = value - (value % disc)
    upper = lower + disc
    return np.where((value - lower) < (upper - value), lower, upper)


def ppf_keys_numpy(xyz, nrm):
    """Keys of all ordered pairs i<j (computePPF.cpp:17-38,91-100), float32 arithmetic as numpy does it.

    Used only to *build* a table for synthetic runs; the table is an input to both the oracle and the
    product, so it needs to be plausible, not bit-equal to any reference output.
    """
    xyz = xyz.astype(np.float32)
    n = nrm.astype(np.float32)
    n = n / np.linalg.norm(n, axis=1, keepdims=True).astype(np.float32)
    m = len(xyz)
    keys = set()
    for i in range(m - 1):
        d = xyz[i + 1:] - xyz[i]  # p2 - p1
        dist = np.linalg.norm(d, axis=1).astype(np.float32)
        ok = dist > 0
        dn = d[ok] / dist[ok, None]
        k0 = (dist[ok] * np.float32(1000)).astype(np.int64)
        a1 = np.degrees(np.arccos(np.clip(dn @ n[i], -1, 1).astype(np.float32)).astype(np.float64)).astype(np.int64)
        a2 = np.degrees(np.arccos(np.clip(np.einsum("ij,ij->i", dn, n[i + 1:][ok]), -1, 1).astype(np.float32)).astype(np.float64)).astype(np.int64)
        a3 = np.degrees(np.arccos(np.clip(n[i + 1:][ok] @ n[i], -1, 1).astype(np.float32)).astype(np.float64)).astype(np.int64)
        kk = np.stack([_closest_bin(k0, _DIST_DISCRET), _closest_bin(a1, _ANGLE_DISCRET),
                       _closest_bin(a2, _ANGLE_DISCRET), _closest_bin(a3, _ANGLE_DISCRET)], axis=1)
        keys.update(map(tuple, np.unique(kk, axis=0).tolist()))
    out = np.array(sorted(keys), dtype=np.int32).reshape(-1, 4)
    return np.ascontiguousarray(out)


def ppf_key_table(semi=SEMI_AXES, density=0.005):
    xyz, nrm = ellipsoid_model_spacing(density, semi)
    return ppf_keys_numpy(xyz, nrm)


# --------------------------------------------------------------------------- replay poses
def replay_poses(gt_pose, h: int, seed: int = 11, max_rot_deg=30.0, max_trans=0.015):
    """``h`` row-major 4x4 float32 poses = gt o perturbation (rot <= 30 deg, trans <= 15 mm)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = np.empty((h, 4, 4), dtype=np.float32)
    for k in range(h):
        axis = rng.normal(size=3)
        ang = math.radians(max_rot_deg) * rng.uniform() ** 2  # denser near the truth
        dR = rot_from_axis_angle(axis, ang)
        dt = rng.normal(size=3)
        dt = dt / np.linalg.norm(dt) * max_trans * rng.uniform() ** 2
        out[k] = (gt_pose @ se3(dR, dt)).astype(np.float32)
    return out


# --------------------------------------------------------------------------- stand-in T42 hand
FINGER_NAMES = ("finger_1_1", "finger_1_2", "finger_2_1", "finger_2_2")
PAIR_NAMES = {"finger_1_1": "finger_2_1", "finger_2_1": "finger_1_1", "finger_1_2": "finger_2_2",
              "finger_2_2": "finger_1_2"}


def _box_cloud(lo, hi, spacing):
    """Points on the surface of an axis-aligned box with outward normals."""
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    pts, nrm = [], []
    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        g0 = np.arange(lo[o[0]], hi[o[0]] + 1e-9, spacing)
        g1 = np.arange(lo[o[1]], hi[o[1]] + 1e-9, spacing)
        u, v = np.meshgrid(g0, g1, indexing="ij")
        for val, sgn in ((lo[ax], -1.0), (hi[ax], 1.0)):
            p = np.zeros((u.size, 3))
            p[:, o[0]] = u.ravel()
            p[:, o[1]] = v.ravel()
            p[:, ax] = val
            n = np.zeros((u.size, 3))
            n[:, ax] = sgn
            pts.append(p)
            nrm.append(n)
    return np.concatenate(pts).astype(np.float32), np.concatenate(nrm).astype(np.float32)


# --------------------------------------------------------------------------- triangle meshes (row N1)
def _orient_outward(V, F, centre=None):
    """Flip triangles whose normal points towards ``centre`` (star-shaped meshes only)."""
    V64 = V.astype(np.float64)
    c = V64.mean(axis=0) if centre is None else np.asarray(centre, dtype=np.float64)
    a, b, d = V64[F[:, 0]], V64[F[:, 1]], V64[F[:, 2]]
    n = np.cross(b - a, d - a)
    flip = np.einsum("ij,ij->i", n, (a + b + d) / 3 - c) < 0
    F = F.copy()
    F[flip] = F[flip][:, [0, 2, 1]]
    return F


def icosphere(subdiv: int = 2):
    t = (1.0 + math.sqrt(5.0)) / 2.0
    V = [(-1, t, 0), (1, t, 0), (-1, -t, 0), (1, -t, 0), (0, -1, t), (0, 1, t), (0, -1, -t), (0, 1, -t),
         (t, 0, -1), (t, 0, 1), (-t, 0, -1), (-t, 0, 1)]
    F = [(0, 11, 5), (0, 5, 1), (0, 1, 7), (0, 7, 10), (0, 10, 11), (1, 5, 9), (5, 11, 4), (11, 10, 2), (10, 7, 6),
         (7, 1, 8), (3, 9, 4), (3, 4, 2), (3, 2, 6), (3, 6, 8), (3, 8, 9), (4, 9, 5), (2, 4, 11), (6, 2, 10),
         (8, 6, 7), (9, 8, 1)]
    V = [np.asarray(v, dtype=np.float64) / math.sqrt(1 + t * t) for v in V]
    for _ in range(subdiv):
        cache, F2 = {}, []

        def mid(i, j):
            key = (min(i, j), max(i, j))
            if key not in cache:
                m = V[i] + V[j]
                V.append(m / np.linalg.norm(m))
                cache[key] = len(V) - 1
            return cache[key]

        for a, b, c in F:
            ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
            F2 += [(a, ab, ca), (b, bc, ab), (c, ca, bc), (ab, bc, ca)]
        F = F2
    return np.asarray(V), np.asarray(F, dtype=np.int32)


def ellipsoid_mesh(semi=SEMI_AXES, subdiv: int = 3):
    """Closed convex mesh of the "ellipse" object (20 * 4**subdiv faces), outward orientation."""
    V, F = icosphere(subdiv)
    V = (V * np.asarray(semi, dtype=np.float64)).astype(np.float32)
    return np.ascontiguousarray(V), np.ascontiguousarray(_orient_outward(V, F, (0, 0, 0)))


def box_mesh(lo, hi, div=(2, 2, 4)):
    """Closed box with ``div`` cells per axis on every side (2 triangles per cell), outward orientation."""
    lo, hi = np.asarray(lo, dtype=np.float64), np.asarray(hi, dtype=np.float64)
    index, V, F = {}, [], []

    def vid(p):
        key = tuple(np.round(p, 9))
        if key not in index:
            index[key] = len(V)
            V.append(p)
        return index[key]

    for ax in range(3):
        o = [a for a in range(3) if a != ax]
        for val in (lo[ax], hi[ax]):
            g0 = np.linspace(lo[o[0]], hi[o[0]], div[o[0]] + 1)
            g1 = np.linspace(lo[o[1]], hi[o[1]], div[o[1]] + 1)
            for i in range(div[o[0]]):
                for j in range(div[o[1]]):
                    q = []
                    for u, v in ((g0[i], g1[j]), (g0[i + 1], g1[j]), (g0[i + 1], g1[j + 1]), (g0[i], g1[j + 1])):
                        p = np.zeros(3)
                        p[o[0]], p[o[1]], p[ax] = u, v, val
                        q.append(vid(p))
                    F += [(q[0], q[1], q[2]), (q[0], q[2], q[3])]
    V = np.asarray(V, dtype=np.float32)
    F = np.asarray(F, dtype=np.int32)
    return np.ascontiguousarray(V), np.ascontiguousarray(_orient_outward(V, F, 0.5 * (lo + hi)))


def torus_mesh(R=0.035, r=0.012, nu=24, nv=12):
    """Closed non-convex mesh (genus 1), outward orientation by construction."""
    V, F = [], []
    for i in range(nu):
        u = 2 * math.pi * i / nu
        for j in range(nv):
            v = 2 * math.pi * j / nv
            V.append(((R + r * math.cos(v)) * math.cos(u), (R + r * math.cos(v)) * math.sin(u), r * math.sin(v)))
    for i in range(nu):
        for j in range(nv):
            a, b = i * nv + j, ((i + 1) % nu) * nv + j
            c, d = ((i + 1) % nu) * nv + (j + 1) % nv, i * nv + (j + 1) % nv
            F += [(a, b, c), (a, c, d)]
    return np.asarray(V, dtype=np.float32), np.asarray(F, dtype=np.int32)


def lshape_mesh(size=0.05, thick=0.02, div=3):
    """Closed non-convex L-shaped prism with a reflex edge, built from an extruded polygon."""
    s, t = size, thick
    poly = [(0, 0), (s, 0), (s, t), (t, t), (t, s), (0, s)]  # counter-clockwise
    n = len(poly)
    zs = np.linspace(-0.5 * t, 0.5 * t, div + 1)
    V = [(x, y, z) for z in zs for (x, y) in poly]
    F = []
    for k in range(div):
        for i in range(n):
            a, b = k * n + i, k * n + (i + 1) % n
            c, d = (k + 1) * n + (i + 1) % n, (k + 1) * n + i
            F += [(a, b, c), (a, c, d)]
    caps = [(0, 1, 2), (0, 2, 3), (0, 3, 4), (0, 4, 5)]
    top = div * n
    for a, b, c in caps:
        F += [(a, c, b), (top + a, top + b, top + c)]
    return np.asarray(V, dtype=np.float32), np.asarray(F, dtype=np.int32)


@dataclass
class HandModel:
    """Kinematic stand-in: names, parents, link->parent transforms and per-link clouds (link frame).

    Mirrors what ``Hand::parseURDF`` leaves behind (Hand.cpp:375-502): ``_clouds``, ``_parent_names``,
    ``_tf_in_parent``; ``_tf_self`` starts as identity.  In the link frame +y is the inner (grasping)
    side and the link extends along -z ... +z, matching how objFuncPSO reads ``FingerProperty``
    (Hand.cpp:24-54,141-152).
    """
    parents: dict = field(default_factory=dict)
    tf_in_parent: dict = field(default_factory=dict)
    clouds: dict = field(default_factory=dict)  # name -> (xyz, nrm)
    meshes: dict = field(default_factory=dict)  # name -> (V, F): Hand::_convex_meshes (Hand.cpp:526-530), link frame


def t42_hand(spacing=0.005) -> HandModel:
    """Stand-in T42: two fingers of two links.  Link frame as the reference uses it (Hand.cpp:24-54):
    the link extends towards -z (tip at min z), +y is the inner (grasping) side, the joint axis is x.
    Hand-base frame: fingers extend towards -x, finger 1 sits at y=-0.04 (inner side +y), finger 2 at
    y=+0.04 (inner side -y), as the crops of main_realdata_auto.cpp:80-93 assume."""
    h = HandModel()
    prox = _box_cloud((-0.010, -0.006, -0.060), (0.010, 0.006, 0.0), spacing)
    dist = _box_cloud((-0.009, -0.005, -0.045), (0.009, 0.005, 0.0), spacing)
    base = _box_cloud((-0.06, -0.05, -0.02), (0.0, 0.05, 0.02), spacing)
    h.clouds["base_link"] = base
    h.parents["base_link"] = "world"
    h.tf_in_parent["base_link"] = np.eye(4, dtype=np.float32)
    R1 = np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], dtype=np.float64)   # link x,y,z -> hand base
    R2 = np.array([[0, 0, 1], [0, -1, 0], [1, 0, 0]], dtype=np.float64)
    f1 = se3(R1, [-0.070, -0.040, 0.0])
    f2 = se3(R2, [-0.070, 0.040, 0.0])
    d = se3(np.eye(3), [0.0, 0.0, -0.062])
    for name, parent, tf, cloud in (
        ("finger_1_1", "base_link", f1, prox), ("finger_1_2", "finger_1_1", d, dist),
        ("finger_2_1", "base_link", f2, prox), ("finger_2_2", "finger_2_1", d, dist),
    ):
        h.parents[name] = parent
        h.tf_in_parent[name] = tf.astype(np.float32)
        h.clouds[name] = cloud
    h.meshes["base_link"] = box_mesh((-0.06, -0.05, -0.02), (0.0, 0.05, 0.02), (3, 5, 2))
    for name in ("finger_1_1", "finger_2_1"):
        h.meshes[name] = box_mesh((-0.010, -0.006, -0.060), (0.010, 0.006, 0.0), (2, 2, 6))
    for name in ("finger_1_2", "finger_2_2"):
        h.meshes[name] = box_mesh((-0.009, -0.005, -0.045), (0.009, 0.005, 0.0), (2, 2, 5))
    return h


def rx(angle):
    c, s = math.cos(angle), math.sin(angle)
    T = np.eye(4)
    T[1, 1], T[1, 2], T[2, 1], T[2, 2] = c, -s, s, c
    return T


def hand_fk(hand: HandModel, angles: dict, name: str):
    """getTFHandBase (Hand.cpp:505-523): link -> hand-base with self rotations Rx(angle)."""
    T = np.eye(4)
    cur = name
    while cur != "base_link":
        T = hand.tf_in_parent[cur].astype(np.float64) @ rx(angles.get(cur, 0.0)) @ T
        cur = hand.parents[cur]
    return T


def make_hand_scene(hand: HandModel, true_angles: dict, n_points: int, seed: int = 5, noise=0.0004,
                    object_frac=0.2):
    """Hand-region scene in the hand-base frame: the posed finger links seen from a camera above the hand
    (+z side), plus a grasped-object blob between the finger tips; resampled to ``n_points``.
    Returns (xyz, nrm) float32, normals towards the camera.  ``scene_remove_swivel`` of the reference is
    the subset with x < -0.1 (Hand.cpp:316-320)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    cam = np.array([-0.12, 0.0, 0.45])
    pts, nrm = [], []
    dense = t42_hand(spacing=0.0011)
    for name in FINGER_NAMES:
        T = hand_fk(hand, true_angles, name)
        x, n = dense.clouds[name]
        xw, nw = apply(T, x).astype(np.float64), rotate(T, n).astype(np.float64)
        view = xw - cam
        view /= np.linalg.norm(view, axis=1, keepdims=True)
        vis = np.einsum("ij,ij->i", nw, view) < -0.15
        pts.append(xw[vis])
        nrm.append(nw[vis])
    pts = np.concatenate(pts)
    nrm = np.concatenate(nrm)
    n_obj = int(n_points * object_frac)
    n_f = n_points - n_obj
    sel = rng.choice(len(pts), size=n_f, replace=len(pts) < n_f)
    pts, nrm = pts[sel] + rng.normal(scale=noise, size=(n_f, 3)), nrm[sel]
    # object: upper half of a small ellipsoid that just fits between the two finger tips
    tips = []
    for name in ("finger_1_2", "finger_2_2"):
        T = hand_fk(hand, true_angles, name)
        x, _ = hand.clouds[name]
        tip_local = np.array([[0.0, x[:, 1].max(), x[:, 2].min()]])
        tips.append(apply(T, tip_local.astype(np.float32))[0].astype(np.float64))
    gap = abs(tips[1][1] - tips[0][1])
    centre = 0.5 * (tips[0] + tips[1])
    semi = (0.020, max(0.5 * gap - 0.001, 0.002), 0.012)
    ox, on = ellipsoid_model(n_obj * 3 + 16, semi)
    up = on[:, 2] > 0.1
    ox, on = ox[up][:n_obj], on[up][:n_obj]
    ox = ox.astype(np.float64) + np.array([centre[0], centre[1], 0.0])
    xyz = np.concatenate([pts, ox]).astype(np.float32)
    nn = np.concatenate([nrm, on.astype(np.float64)]).astype(np.float32)
    order = rng.permutation(len(xyz))
    return np.ascontiguousarray(xyz[order]), np.ascontiguousarray(nn[order])


# --------------------------------------------------------------------------- row N1: physics rejection
FINGER_ORDER = ("finger_1_1", "finger_1_2", "finger_2_1", "finger_2_2")


def physics_case(n_hyp: int = 64, seed: int = 3, n_model: int = 400, n_scene: int = 3000, mesh_subdiv: int = 2,
                 finger_status=(1, 1, 1, 1), spacing=0.005, max_rot_deg=25.0, max_trans=0.02, finger_angles=None):
    """Inputs of PoseEstimator::rejectByCollisionOrNonTouching (PoseEstimator.cpp:524-735) for a synthetic grasp.

    The ellipsoid sits between the two fingers of the stand-in hand (its 40 mm semi-axis across the 68 mm gap, so the
    true pose touches both sides); the hypotheses are the true pose under perturbations of growing size plus a few
    targeted ones (pushed into a finger, pulled out of the hand, shifted along the fingers).
    Returns (p, poses): ``p`` is the dict ``Context.physics_set_frame`` / ``orc.physics_args`` take (the meshes are
    given both as arrays for the oracle and, under ``meshes``, as (id, V, F, pose) tuples to register)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    hand = t42_hand(spacing)
    angles = {"finger_1_1": 0.0, "finger_1_2": 0.0, "finger_2_1": 0.0, "finger_2_2": 0.0}
    angles.update(finger_angles or {})
    Rc = rot_from_axis_angle([1.0, 0.2, -0.1], 2.6)
    handbase_in_cam = se3(Rc, [0.03, -0.02, 0.55])
    R_obj = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.float64)  # model x (40 mm) -> hand y
    obj_in_hand = se3(R_obj, [-0.150, 0.0, 0.0])
    gt = handbase_in_cam @ obj_in_hand
    mx, mn_ = ellipsoid_model(n_model)
    V, F = ellipsoid_mesh(subdiv=mesh_subdiv)
    ext = mx.max(0) - mx.min(0)
    # scene without the hand (camera frame): the camera-facing part of the object plus clutter behind it
    sx, sn = ellipsoid_model(n_scene * 2)
    sw, nw = apply(gt, sx).astype(np.float64), rotate(gt, sn).astype(np.float64)
    vis = np.einsum("ij,ij->i", nw, sw / np.linalg.norm(sw, axis=1, keepdims=True)) < -0.05
    obj_pts = sw[vis][: int(n_scene * 0.8)] + rng.normal(scale=0.0004, size=(min(int(n_scene * 0.8), int(vis.sum())), 3))
    clutter = gt[:3, 3] + rng.normal(scale=0.12, size=(n_scene - len(obj_pts), 3)) + np.array([0, 0, 0.15])
    local = (clutter - gt[:3, 3]) @ gt[:3, :3]  # a depth camera sees no point inside (or hugging) the object
    clutter = clutter[((local / (np.asarray(SEMI_AXES) + 0.02)) ** 2).sum(axis=1) > 1.0]
    cloud_without_hand = np.concatenate([obj_pts, clutter]).astype(np.float32)
    cloud_without_hand = np.ascontiguousarray(cloud_without_hand[rng.permutation(len(cloud_without_hand))])
    f2h = [hand_fk(hand, angles, name).astype(np.float32) for name in FINGER_ORDER]
    hand_cloud = np.concatenate([apply(hand_fk(hand, angles, name) if name != "base_link" else np.eye(4), hand.clouds[name][0])
                                 for name in ("base_link",) + FINGER_ORDER]).astype(np.float32)
    # hypotheses
    poses = list(replay_poses(gt, max(n_hyp - 8, 0), seed=seed + 1, max_rot_deg=max_rot_deg, max_trans=max_trans))

    def moved(dx, dy, dz):
        return handbase_in_cam @ se3(np.eye(3), [dx, dy, dz]) @ obj_in_hand

    poses += [gt, moved(0, 0.030, 0), moved(0, -0.028, 0), moved(-0.12, 0, 0), moved(0.02, 0, 0.05), moved(0.06, 0.0, 0.0),
              moved(0, 0, -0.012), moved(-0.02, 0.004, 0.0)]
    poses = np.ascontiguousarray(np.asarray(poses[-n_hyp:] if n_hyp < len(poses) else poses, dtype=np.float32))
    p = dict(
        object_V=V, object_F=F,
        finger_V=[hand.meshes[n][0] for n in FINGER_ORDER], finger_F=[hand.meshes[n][1] for n in FINGER_ORDER],
        finger_mesh_pose=f2h,
        finger_xyz=[hand.clouds[n][0] for n in FINGER_ORDER], finger2handbase=f2h, finger_status=list(finger_status),
        hand_cloud=hand_cloud, cloud_without_hand=cloud_without_hand,
        cam2handbase=np.linalg.inv(handbase_in_cam).astype(np.float32),
        model=mx, model_center_init=mx.mean(0).astype(np.float32),
        smallest_dim=float(ext.min()), ob_diameter=float(np.linalg.norm(ext)),
        collision_thres=0.4, non_touch_dist=0.01, collision_finger_dist=0.012, collision_finger_volume_ratio=0.25,  # config_autodataset.yaml:128-131
        voxel_size=0.005,
        object_mesh=0, finger_mesh=[1, 2, 3, 4],
    )
    p["meshes"] = [(0, V, F, None)] + [(1 + k, p["finger_V"][k], p["finger_F"][k], f2h[k]) for k in range(4)]
    return p, poses


# --------------------------------------------------------------------------- a whole frame: hand + grasped object, one camera
def _look_at(eye, target, up=(0.0, 1.0, 0.0)):
    """Camera pose (camera -> world) with +z towards the target, +x right, +y down (OpenCV convention)."""
    eye, target = np.asarray(eye, np.float64), np.asarray(target, np.float64)
    z = target - eye
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    T = np.eye(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, eye
    return T


def grasp_frame(seed: int = 2, n_object: int = 9000, hand_spacing=0.0013, noise=0.0003):
    """One synthetic frame of the whole pipeline (main_realdata_auto.cpp:54-205): the stand-in hand at given finger
    angles holding the ellipsoid, seen by one camera.  Returns a dict with the dense scene in the camera frame (points and
    normals towards the camera, as the integral-image estimator leaves them), the true hand-base pose and a perturbed
    one (what the robot reports), the object's true pose, the hand model with its angles, and the meshes."""
    rng = np.random.Generator(np.random.PCG64(seed))
    hand = t42_hand()
    angles = {"finger_1_1": math.radians(4), "finger_1_2": math.radians(3), "finger_2_1": math.radians(5), "finger_2_2": math.radians(2)}
    cam_in_handbase = _look_at([-0.34, 0.22, 0.30], [-0.09, 0.0, 0.0], up=(0.0, 0.0, -1.0))  # oblique: top and two sides of every link
    handbase_in_cam = np.linalg.inv(cam_in_handbase)
    eye = cam_in_handbase[:3, 3]
    R_obj = np.array([[0, 0, 1], [1, 0, 0], [0, 1, 0]], dtype=np.float64)  # model x (40 mm) across the fingers
    obj_in_hand = se3(R_obj, [-0.150, 0.0, 0.0])
    pts, nrm, is_obj = [], [], []
    dense = t42_hand(spacing=hand_spacing)
    for name in dense.clouds:
        Tl = np.eye(4) if name == "base_link" else hand_fk(hand, angles, name)
        x, n = apply(Tl, dense.clouds[name][0]).astype(np.float64), rotate(Tl, dense.clouds[name][1]).astype(np.float64)
        view = x - eye
        view /= np.linalg.norm(view, axis=1, keepdims=True)
        vis = np.einsum("ij,ij->i", n, view) < -0.15
        pts.append(x[vis]), nrm.append(n[vis]), is_obj.append(np.zeros(int(vis.sum()), bool))
    ox, on = ellipsoid_model(int(n_object * 2.6))
    x, n = apply(obj_in_hand, ox).astype(np.float64), rotate(obj_in_hand, on).astype(np.float64)
    view = x - eye
    view /= np.linalg.norm(view, axis=1, keepdims=True)
    vis = np.einsum("ij,ij->i", n, view) < -0.1
    # the fingers hide what lies behind them from this camera: drop object points inside a finger box footprint
    for name in FINGER_NAMES:
        Tl = np.linalg.inv(hand_fk(hand, angles, name))
        loc = x @ Tl[:3, :3].T + Tl[:3, 3]
        lo, hi = hand.clouds[name][0].min(0) - 0.001, hand.clouds[name][0].max(0) + 0.001
        vis &= ~((loc >= lo) & (loc <= hi)).all(axis=1)
    pts.append(x[vis][:n_object]), nrm.append(n[vis][:n_object]), is_obj.append(np.ones(min(int(vis.sum()), n_object), bool))
    pts, nrm, is_obj = np.concatenate(pts), np.concatenate(nrm), np.concatenate(is_obj)
    pts = pts + rng.normal(scale=noise, size=pts.shape)
    order = rng.permutation(len(pts))
    pts, nrm, is_obj = pts[order], nrm[order], is_obj[order]
    D = se3(rot_from_axis_angle([0.3, 1.0, 0.2], math.radians(2.0)), [0.003, -0.002, 0.002])
    V, F = ellipsoid_mesh(subdiv=3)
    return dict(scene_xyz=apply(handbase_in_cam, pts.astype(np.float32)), scene_nrm=rotate(handbase_in_cam, nrm.astype(np.float32)), is_object=is_obj,
                handbase_in_cam=handbase_in_cam.astype(np.float32), handbase_in_cam_reported=(handbase_in_cam @ D).astype(np.float32),
                object_in_cam=(handbase_in_cam @ obj_in_hand).astype(np.float32), hand=hand, angles=angles, object_V=V, object_F=F)


# --------------------------------------------------------------------------- depth images (rows N2 / N3 / N4)
def hand_mesh_in_cam(hand: HandModel, angles: dict, handbase_in_cam, names=None):
    """The triangles of the hand's link meshes moved by handbase_in_cam * getTFHandBase(name): what
    PoseEstimator::rejectByRender adds to the renderer for every matched component (PoseEstimator.cpp:362-383)."""
    Vs, Fs, off = [], [], 0
    for name in (names if names is not None else hand.meshes):
        V, F = hand.meshes[name]
        T = np.asarray(handbase_in_cam, np.float64) @ (np.eye(4) if name == "base_link" else hand_fk(hand, angles, name))
        Vs.append(apply(T, np.asarray(V, np.float32)).astype(np.float32))
        Fs.append(np.asarray(F, np.int32) + off)
        off += len(V)
    if not Vs:
        return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)
    return np.concatenate(Vs), np.concatenate(Fs)


def render_depth_numpy(V, F, K, H, W):
    """Nearest eye depth (metres, 0 = nothing) per pixel of a triangle soup given in the camera frame: pinhole
    u = fx X / Z + cx, v = fy Y / Z + cy sampled at integer pixels (what a depth sensor image means to
    Utils::convert3dOrganizedRGB).  A plain z-buffer for the synthetic frames; not the reference's renderer (that is
    hop_render_depth, with OpenGL's half-pixel and mirrored principal point)."""
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    V = np.asarray(V, np.float64)
    z = np.full((H, W), np.inf)
    P = V[np.asarray(F)]                                   # (nf, 3, 3)
    ok = (P[:, :, 2] > 1e-6).all(axis=1)
    x = fx * P[:, :, 0] / np.where(P[:, :, 2] > 1e-6, P[:, :, 2], 1.0) + cx
    y = fy * P[:, :, 1] / np.where(P[:, :, 2] > 1e-6, P[:, :, 2], 1.0) + cy
    iz = 1.0 / np.where(P[:, :, 2] > 1e-6, P[:, :, 2], 1.0)
    for f in np.flatnonzero(ok):
        w0, w1 = max(0, int(math.ceil(x[f].min()))), min(W - 1, int(math.floor(x[f].max())))
        h0, h1 = max(0, int(math.ceil(y[f].min()))), min(H - 1, int(math.floor(y[f].max())))
        if w0 > w1 or h0 > h1:
            continue
        px, py = np.meshgrid(np.arange(w0, w1 + 1, dtype=np.float64), np.arange(h0, h1 + 1, dtype=np.float64))
        area = (x[f, 1] - x[f, 0]) * (y[f, 2] - y[f, 0]) - (x[f, 2] - x[f, 0]) * (y[f, 1] - y[f, 0])
        if area == 0:
            continue
        e0 = (x[f, 2] - x[f, 1]) * (py - y[f, 1]) - (y[f, 2] - y[f, 1]) * (px - x[f, 1])
        e1 = (x[f, 0] - x[f, 2]) * (py - y[f, 2]) - (y[f, 0] - y[f, 2]) * (px - x[f, 2])
        e2 = (x[f, 1] - x[f, 0]) * (py - y[f, 0]) - (y[f, 1] - y[f, 0]) * (px - x[f, 0])
        inside = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
        if not inside.any():
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            Z = area / (e0 * iz[f, 0] + e1 * iz[f, 1] + e2 * iz[f, 2])
        sub = z[h0:h1 + 1, w0:w1 + 1]
        upd = inside & (Z > 0) & (Z < sub)
        sub[upd] = Z[upd]
    z[~np.isfinite(z)] = 0.0
    return z


CAM_K = np.array([[615.0, 0.0, 320.0], [0.0, 615.0, 240.0], [0.0, 0.0, 1.0]], dtype=np.float32)  # SR300-like, 640 x 480


def grasp_depth_frame(seed: int = 2, noise_mm: float = 0.3, table: bool = True):
    """The synthetic grasp of ``grasp_frame`` as what the robot records (run_real_all.cpp:72-104): a 16-bit depth image in
    millimetres (hand + object meshes, optionally a table plane behind them, rendered through ``CAM_K``, Gaussian depth
    noise), the camera intrinsics, the true and the reported hand-base pose, the object's pose, the hand model with its
    finger angles, and the meshes."""
    g = grasp_frame(seed=seed, n_object=200)
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    hand = g["hand"]
    hV, hF = hand_mesh_in_cam(hand, g["angles"], g["handbase_in_cam"])
    oV = apply(g["object_in_cam"], np.asarray(g["object_V"], np.float32))
    V = np.concatenate([hV, oV])
    F = np.concatenate([hF, np.asarray(g["object_F"], np.int32) + len(hV)])
    if table:   # a plane 12 cm behind the hand, facing the camera: the background a real frame has
        zc = float(np.concatenate([hV, oV])[:, 2].max()) + 0.12
        q = np.array([[-0.6, -0.5, zc], [0.6, -0.5, zc], [0.6, 0.5, zc], [-0.6, 0.5, zc]], np.float32)
        V = np.concatenate([V, q])
        F = np.concatenate([F, np.array([[0, 1, 2], [0, 2, 3]], np.int32) + len(V) - 4])
    H, W = 480, 640
    z = render_depth_numpy(V, F, CAM_K, H, W)
    mm = z * 1000.0 + np.where(z > 0, rng.normal(0.0, noise_mm, z.shape), 0.0)
    depth = np.clip(np.rint(mm), 0, 65535).astype(np.uint16)
    out = dict(g)
    out.update(depth=depth, K=CAM_K.copy(), hand_mesh_cam=(hV, hF))
    return out
