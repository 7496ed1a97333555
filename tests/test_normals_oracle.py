"""CPU tests of the oracle's restatement of the two PCL normal estimators (PCL 1.9 is absent: "parity
unpinned") on cases whose answer can be computed by hand."""
import numpy as np


def _organised_plane(H=60, W=80, fx=300.0, normal=(0.2, -0.1, 1.0), d0=0.5):
    """Pixels back-projected onto the plane n . p = n_z * d0 (depth of the ray through every pixel)."""
    n = np.asarray(normal, np.float64)
    n /= np.linalg.norm(n)
    v, u = np.meshgrid(np.arange(W), np.arange(H))
    rx, ry = (v - W / 2) / fx, (u - H / 2) / fx
    z = n[2] * d0 / (n[0] * rx + n[1] * ry + n[2])
    return np.stack([rx * z, ry * z, z], axis=-1).astype(np.float32), n


def test_integral_image_normals_on_a_plane(orc):
    xyz, n = _organised_plane()
    nrm = orc.normals_integral_image(xyz, 0.02, 10.0, True)
    H, W = xyz.shape[:2]
    # border of normal_smoothing_size pixels undefined (BORDER_POLICY_IGNORE)
    assert np.isnan(nrm[:10]).all() and np.isnan(nrm[-10:]).all() and np.isnan(nrm[:, :10]).all() and np.isnan(nrm[:, -10:]).all()
    inner = nrm[10:-10, 10:-10].reshape(-1, 3)
    assert np.isfinite(inner).all()
    # every gradient lies in the plane, so the normal is the plane's, flipped towards the origin: -n (n_z > 0, plane in front)
    assert np.abs(inner + n.astype(np.float32)).max() < 2e-4
    assert np.abs(np.linalg.norm(inner, axis=1) - 1).max() < 1e-6


def test_integral_image_normals_depth_edge_and_dropped_pixels(orc):
    xyz, n = _organised_plane(H=64, W=96)
    xyz[:, 48:, 2] += 0.5          # a 0.5 m step: > 0.02 * (0.5 + 1) * 2
    xyz[:, 48:, :2] *= (xyz[:, 48:, 2:3] / (xyz[:, 48:, 2:3] - 0.5))
    xyz[30:34, 20:24] = 0.0        # dropped pixels: the reference's bad_point (0,0,0), a finite point for PCL
    nrm = orc.normals_integral_image(xyz, 0.02, 10.0, True)
    # distance map <= 2 next to the step: undefined there, defined 6 pixels away on both sides
    assert np.isnan(nrm[20, 46:50]).all()
    assert np.isfinite(nrm[20, 40]).all() and np.isfinite(nrm[20, 56]).all()
    assert np.isnan(nrm[31, 21]).all() and np.isnan(nrm[31, 25]).all()
    # NaN input is read as a dropped pixel
    xyz2 = xyz.copy()
    xyz2[30:34, 20:24] = np.nan
    nrm2 = orc.normals_integral_image(xyz2, 0.02, 10.0, True)
    assert np.array_equal(np.isnan(nrm), np.isnan(nrm2)) and np.allclose(np.nan_to_num(nrm), np.nan_to_num(nrm2))


def test_mls_on_a_plane_and_isolated_points(orc):
    g = np.arange(-12, 13) * 0.001
    X, Y = np.meshgrid(g, g)
    R = np.array([[0.936, 0.0, 0.352], [0.0, 1.0, 0.0], [-0.352, 0.0, 0.936]])
    plane = (np.stack([X.ravel(), Y.ravel(), np.zeros(X.size)], axis=1) @ R.T + np.array([0.05, -0.02, 0.4])).astype(np.float32)
    lonely = np.array([[0.5, 0.5, 0.5], [0.5, 0.5003, 0.5]], np.float32)   # 2 neighbours each (< 3): dropped
    xyz = np.vstack([plane, lonely])
    p, n, curv, idx = orc.normals_mls(xyz, 0.003, 2)
    assert len(p) == len(plane) and np.array_equal(idx, np.arange(len(plane)))
    nz = R[:, 2]
    assert np.abs(np.abs(n @ nz) - 1).max() < 1e-5          # the plane's normal (sign: PCL's eigenvector, unflipped)
    assert np.abs(p - plane).max() < 2e-6                   # points already on the surface stay
    assert curv.max() < 1e-4


def test_mls_on_a_sphere_projects_onto_it(orc):
    rng = np.random.default_rng(3)
    r = 0.04
    d = rng.normal(size=(20000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d[d[:, 2] > 0.8]                                    # a cap, ~1 mm spacing
    noise = rng.normal(0, 0.0002, size=len(d))
    xyz = (d * (r + noise[:, None])).astype(np.float32)
    p, n, curv, idx = orc.normals_mls(xyz, 0.003, 2)
    inner = d[idx][:, 2] > 0.85
    rad = np.linalg.norm(p[inner].astype(np.float64), axis=1)
    assert np.abs(rad - r).std() < 0.6 * np.abs(noise).std()            # smoothing: closer to the sphere than the input
    cosang = np.abs(np.sum(n[inner] * d[idx][inner], axis=1))
    assert np.degrees(np.arccos(np.clip(cosang, -1, 1))).mean() < 3.0


def test_scene_from_depth_normals_on_the_example_frame(orc, golden_dir):
    import os
    g = np.load(os.path.join(golden_dir, "depth7_raw.npz"))
    lo, hi = (-0.25, -0.2, -0.12), (-0.07, 0.2, 0.05)
    xyz, nrm = orc.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    ref, counts = orc.scene_from_depth(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    assert len(xyz) == counts[2] and np.abs(xyz - ref).max() < 1e-6
    ok = np.isfinite(nrm).all(axis=1)
    assert ok.mean() > 0.5
    assert np.abs(np.linalg.norm(nrm[ok], axis=1) - 1).max() < 1e-3
    assert (np.sum(nrm[ok] * xyz[ok], axis=1) < 0).mean() > 0.99      # towards the camera
