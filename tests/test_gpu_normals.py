"""GPU parity tests of the two normal estimators (Utils::calNormalIntegralImage, Utils::calNormalMLS)
against the CPU oracle.  PCL itself is absent ("parity unpinned"): the oracle is checked on hand-computed cases in
tests/test_normals_oracle.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def api(hop):
    from hop_amd import api as _api
    _api.lib()
    return _api


@pytest.fixture(scope="module")
def ctx(api):
    c = api.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def synth(hop):
    return hop.synth


def _organised(depth_raw, K, unit=0.001):
    """Utils::readDepthImage + convert3dOrganizedRGB (Utils.cpp:36-55,79-115): dropped pixels are (0,0,0)."""
    d = (depth_raw.astype(np.float32).astype(np.float64) * unit).astype(np.float32)
    d[(d > 2.0) | (d < 0.1)] = 0
    H, W = d.shape
    v, u = np.meshgrid(np.arange(W, dtype=np.float32), np.arange(H, dtype=np.float32))
    ok = (d > 0.1) & (d < 2.0)
    x = ((v - K[0, 2]) * d / K[0, 0]).astype(np.float32)
    y = ((u - K[1, 2]) * d / K[1, 1]).astype(np.float32)
    return np.where(ok[..., None], np.stack([x, y, d], axis=-1), 0).astype(np.float32)


def _cmp_normals(a, b, tol=2e-5):
    na, nb = np.isnan(a).any(axis=-1), np.isnan(b).any(axis=-1)
    assert np.array_equal(na, nb), (na.sum(), nb.sum(), (na != nb).sum())
    assert (~na).sum() > 0
    assert np.abs(a[~na] - b[~na]).max() < tol


def test_integral_image_normals_on_the_example_frame(ctx, orc, golden_dir):
    """640 x 480, the reference's example/depth7.png: same undefined pixels (the distance map and its int() decide them)
    and the same normals as the oracle."""
    g = np.load(os.path.join(golden_dir, "depth7_raw.npz"))
    xyz = _organised(g["depth"], g["K"])
    got = ctx.normals_integral_image(xyz, 0.02, 10.0, True)
    ref = orc.normals_integral_image(xyz, 0.02, 10.0, True)
    _cmp_normals(got, ref)
    assert np.isfinite(ref).all(axis=-1).sum() > 30000


@pytest.mark.parametrize("shape,depth_dependent,smoothing", [((64, 96), True, 10.0), ((37, 53), False, 5.0), ((130, 70), True, 7.0)])
def test_integral_image_normals_synthetic(ctx, orc, shape, depth_dependent, smoothing):
    """planes with steps, dropped pixels (0 and NaN), noise; odd sizes; both smoothing modes"""
    H, W = shape
    rng = np.random.default_rng(H * 1000 + W)
    v, u = np.meshgrid(np.arange(W), np.arange(H))
    z = 0.4 + 0.001 * v + 0.0005 * u + rng.normal(0, 0.0003, (H, W))
    z[:, W // 2:] += 0.3
    z[H // 3: H // 3 + 4, 5:20] = 0
    xyz = np.stack([(v - W / 2) * z / 300.0, (u - H / 2) * z / 300.0, z], axis=-1).astype(np.float32)
    xyz[z == 0] = 0
    xyz[H // 2, W // 4] = np.nan
    got = ctx.normals_integral_image(xyz, 0.02, smoothing, depth_dependent)
    ref = orc.normals_integral_image(xyz, 0.02, smoothing, depth_dependent)
    _cmp_normals(got, ref)


def test_mls_matches_oracle(ctx, orc, synth):
    """a noisy 1 mm cloud of the ellipsoid plus a few stragglers: same survivors, projected points <= 1e-6 m, normals
    <= 2e-5 (double-precision fit, neighbour sums in a different order)"""
    mx, mn = synth.ellipsoid_model(12000)
    rng = np.random.default_rng(5)
    xyz = (mx + rng.normal(0, 0.0002, mx.shape)).astype(np.float32)
    xyz = np.vstack([xyz, np.array([[0.3, 0.3, 0.3], [0.3, 0.3005, 0.3], [0.31, 0.3, 0.3]], np.float32)])
    p, n, cv, idx = ctx.normals_mls(xyz, 0.003, 2)
    rp, rn, rcv, ridx = orc.normals_mls(xyz, 0.003, 2)
    assert np.array_equal(idx, ridx) and len(idx) == len(mx)
    assert np.abs(p - rp).max() < 1e-6
    assert np.abs(n - rn).max() < 2e-5
    assert np.abs(cv - rcv).max() < 1e-5
    # and the normals are the ellipsoid's (up to sign) within a few degrees
    c = np.abs(np.sum(n * mn[idx], axis=1))
    assert np.degrees(np.arccos(np.clip(c, 0, 1))).mean() < 4.0


def test_mls_edge_cases(ctx, orc):
    empty = np.zeros((0, 3), np.float32)
    p, n, cv, idx = ctx.normals_mls(empty)
    assert len(p) == 0
    two = np.array([[0, 0, 0.5], [0.001, 0, 0.5]], np.float32)
    assert len(ctx.normals_mls(two)[0]) == 0                  # fewer than 3 neighbours: no output (mls.hpp performProcessing)
    # 3..5 neighbours: plane only (nr_coeff = 6 not reached); collinear points: still an output, as in PCL
    five = np.array([[0, 0, 0.5], [0.001, 0, 0.5], [0, 0.001, 0.5], [0.001, 0.001, 0.5002], [0.0005, 0.0005, 0.5001]], np.float32)
    p, n, cv, idx = ctx.normals_mls(five)
    rp, rn, rcv, ridx = orc.normals_mls(five)
    assert np.array_equal(idx, ridx) and len(idx) == 5
    assert np.abs(p - rp).max() < 1e-6 and np.abs(n - rn).max() < 2e-5


def test_scene_from_depth_normals_on_the_example_frame(ctx, orc, golden_dir):
    """main_realdata_auto.cpp:54-96 with the normals of :61 on example/depth7.png: the cloud Hand::setCurScene receives"""
    g = np.load(os.path.join(golden_dir, "depth7_raw.npz"))
    lo, hi = (-0.25, -0.2, -0.12), (-0.07, 0.2, 0.05)
    xyz, nrm, counts = ctx.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    rx, rn = orc.scene_from_depth_normals(g["depth"], 0.001, g["K"], g["cam_in_handbase"], g["handbase_in_cam"], 0.001, lo, hi)
    assert counts[0] == 68600 and xyz.shape == rx.shape and len(xyz) > 5000
    assert np.abs(xyz - rx).max() < 1e-6
    _cmp_normals(nrm, rn, tol=5e-5)
