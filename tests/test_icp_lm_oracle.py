"""Row S2 (refineByICP / Utils::runICP) and the Eigen calls of row S1 (clusterPoses), PINNED.

The reference's ICP minimiser is PCL's TransformationEstimationPointToPlane = Eigen::LevenbergMarquardt<NumericalDiff<..>, float>
(Utils.cpp:200-216).  Eigen's NonLinearOptimization / NumericalDiff modules are vendored in the reference
(src/OpenGR_4pcs/3rdparty/Eigen/unsupported); oracle/ref_icp_driver.cpp compiles them in place (oracle/_ref/libref_icp.so) and
oracle/gen_golden.py stored what they return: tests/golden/icp_lm_{kat,c1,c2sub}.npz.  CPU tests:
  * every pure function the path evaluates is bit-equal to Eigen's: the warp matrix of a parameter vector, the residual vector,
    the forward-difference Jacobian NumericalDiff hands the minimiser; eulerAngles(2,1,0), rotationGeodesicDistance, (t0-t1).norm(),
    the 4x4 product -- for the oracle AND for the product's host functions (hop_cluster_pose_terms);
  * the restated minimiser (oracle lm_*, which the GPU's nn_mode 5 computes) reaches the same minimum: same cost within the
    minimiser's own ftol, parameters within 5e-4 on well-conditioned sets.  On the ellipse's weakly constrained slides Eigen's float
    run is itself only reproducible to ~1e-3 (its forward-difference Jacobian at |x| ~ 1e-3 has ~10 % noise per entry), which the
    last test quantifies with a SECOND BUILD of the reference (-march=native): the restatement is as close to the reference as the
    reference's other build is.
"""
import math
import os

import numpy as np
import pytest


def biteq(a, b):
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return a.shape == b.shape and np.array_equal(a.view(np.int32), b.view(np.int32))


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "icp_lm_kat.npz"))


def test_warp_matrix_is_bit_equal_to_eigen(orc, kat):
    W = np.stack([orc.lm_warp6(x) for x in kat["warp_x"]])
    assert biteq(W, kat["warp_T"])


def test_residuals_and_numerical_jacobian_are_bit_equal_to_eigen(orc, kat):
    for k in range(int(kat["n_lm_sets"])):
        P, Q, N = kat[f"lm{k}_P"], kat[f"lm{k}_Q"], kat[f"lm{k}_N"]
        for x, f, J in zip(kat[f"lm{k}_probe_x"], kat[f"lm{k}_probe_f"], kat[f"lm{k}_probe_J"]):
            f1, J1 = orc.lm_residuals_jacobian(P, Q, N, x)
            assert biteq(f1, f), k
            assert biteq(J1, J), k


def test_euler_geodesic_translation_and_product_are_bit_equal_to_eigen(orc, kat):
    """clusterPoses' three comparisons (PoseEstimator.cpp:148-196, Utils.cpp:29-32) and ICP's final = T * final."""
    E = np.stack([orc.euler_zyx(r) for r in kat["euler_R"]])
    assert biteq(E, kat["euler_zyx"])
    G = np.array([orc.geodesic(a, b) for a, b in zip(kat["geo_R1"], kat["geo_R2"])], np.float32)
    assert biteq(G, kat["geodesic"])
    Tn = np.array([orc.tdiff_norm(a, b) for a, b in zip(kat["tdiff_t0"], kat["tdiff_t1"])], np.float32)
    assert biteq(Tn, kat["tdiff_norm"])
    M = np.stack([orc.mul4(a, b) for a, b in zip(kat["mat_A"], kat["mat_B"])])
    assert biteq(M, kat["mat_mul"])
    # transformation.inverse() * pose (PoseEstimator.cpp:267): Eigen's SSE 4x4 float inverse vs the adjugate in double here
    Iv = np.stack([orc.inverse_times(a, b) for a, b in zip(kat["mat_A"], kat["mat_B"])])
    assert np.abs(Iv - kat["mat_inv_times"]).max() < 5e-7


def test_product_host_functions_are_bit_equal_to_eigen(hop, kat):
    """the same known answers through libhop.so's host side (hop_cluster_pose_terms: what hop_cluster_poses compares)"""
    from hop_amd import api
    n = 600
    Ra, Rb = kat["geo_R1"][:n], kat["geo_R2"][:n]
    ta, tb = kat["tdiff_t0"][:n], kat["tdiff_t1"][:n]
    for i in range(n):
        A, B = np.eye(4, dtype=np.float32), np.eye(4, dtype=np.float32)
        A[:3, :3], B[:3, :3], A[:3, 3], B[:3, 3] = Ra[i], Rb[i], ta[i], tb[i]
        out = api.cluster_pose_terms(A, B)
        assert biteq(out[:3], kat["euler_zyx"][i]), i
        assert biteq(out[3:4], kat["geodesic"][i:i + 1]), i
        assert biteq(out[4:5], kat["tdiff_norm"][i:i + 1]), i


def _fnorm(orc, P, Q, N, x):
    f, _ = orc.lm_residuals_jacobian(P, Q, N, x)
    return float(np.sqrt((f.astype(np.float64) ** 2).sum()))


def test_restated_minimiser_reaches_eigens_minimum(orc, kat):
    for k in range(int(kat["n_lm_sets"])):
        P, Q, N = kat[f"lm{k}_P"], kat[f"lm{k}_Q"], kat[f"lm{k}_N"]
        T, x, st = orc.lm_point_to_plane(P, Q, N)
        xg, Tg = kat[f"lm{k}_x"], kat[f"lm{k}_T"]
        assert st[0] in (1, 2, 3), (k, st)  # stopped on its tolerances, like Eigen's run
        assert biteq(T, orc.lm_warp6(x))
        c_mine, c_gold, c_start = _fnorm(orc, P, Q, N, x), _fnorm(orc, P, Q, N, xg), _fnorm(orc, P, Q, N, np.zeros(6, np.float32))
        assert c_mine < c_start
        assert abs(c_mine - c_gold) / c_gold < 5e-3, (k, c_mine, c_gold)  # same minimum within the stopping tolerance (measured <= 1.7e-3)
        if k >= 8:  # the well-conditioned sets: the parameters themselves (measured <= 1.4e-4)
            assert np.abs(x - xg).max() < 5e-4, (k, np.abs(x - xg).max())
        else:       # the ellipse: a flat valley along the surface's slides (measured <= 7.3e-3)
            assert np.abs(x - xg).max() < 3e-2, (k, np.abs(x - xg).max())


def test_fewer_than_four_correspondences_leave_the_matrix_untouched(orc):
    P = np.random.default_rng(0).standard_normal((3, 3)).astype(np.float32)
    assert orc.lm_point_to_plane(P, P, P)[0] is None
    if orc.ref_icp_available():
        assert orc.lm_point_to_plane(P, P, P, ref=True)[0] is None


def _rot_deg(Ra, Rb):
    c = (np.trace(Ra.astype(np.float64).T @ Rb.astype(np.float64)) - 1) / 2
    return math.degrees(math.acos(max(-1.0, min(1.0, float(c)))))


def pose_deltas(p, q):
    t = np.array([1e3 * np.linalg.norm(a[:3, 3] - b[:3, 3]) for a, b in zip(p, q)])
    r = np.array([_rot_deg(a[:3, :3], b[:3, :3]) for a, b in zip(p, q)])
    return t, r


def c1_inputs(hop, golden_dir):
    g = np.load(os.path.join(golden_dir, "depth7_hand_region.npz"))
    mx5, mn5 = hop.synth.ellipsoid_model_spacing(0.005)
    return g["xyz"], g["nrm"], mx5, mn5


def c2sub_inputs(hop):
    sc = hop.synth.make_scene(4000, seed=7)
    keep = sc.conf >= 0.8
    mx5, mn5 = hop.synth.ellipsoid_model_spacing(0.005)
    return sc.xyz[keep], sc.nrm[keep], mx5, mn5


def closeness(p, gold):
    """(hypotheses within 1 mm / 1 degree of the golden pose, median translation [mm], median rotation [deg])"""
    t, r = pose_deltas(p, gold)
    return int(((t < 1) & (r < 1)).sum()), float(np.median(t)), float(np.median(r))


def assert_as_close_as_the_other_build(p, g, slack):
    """p is as close to the reference's default build (poses_out) as the reference's -march=native build is (poses_out_native)"""
    n_p, t_p, r_p = closeness(p, g["poses_out"])
    n_n, t_n, r_n = closeness(g["poses_out_native"], g["poses_out"])
    assert n_p >= n_n - slack, (n_p, n_n)
    assert t_p <= 1.5 * t_n + 0.02 and r_p <= 1.5 * r_n + 0.05, ((t_p, r_p), (t_n, r_n))
    return (n_p, t_p, r_p), (n_n, t_n, r_n)


def test_icp_with_the_restated_minimiser_vs_the_reference_builds_c2sub(orc, hop, golden_dir):
    """96 replay poses on a C2-style scene: 93 of 96 within 1 mm / 1 degree of Eigen's run, the reference's native build: 92."""
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    S, Sn, mx5, mn5 = c2sub_inputs(hop)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01)
    mine, native = assert_as_close_as_the_other_build(p, g, slack=3)
    assert mine[0] >= 88
    assert (it == g["iterations"]).sum() >= 88 and np.array_equal(cv, g["converged"])
    # the one-step Gauss-Newton form (nn_mode 0-4) is measurably farther from the reference than its own second build
    pg, _, _ = orc.icp_refine_batch(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01)
    assert closeness(pg, g["poses_out"])[0] < mine[0]


def test_icp_with_the_restated_minimiser_vs_the_reference_builds_c1(orc, hop, golden_dir):
    """The C1 frame (example/depth7.png: a hand, no ellipse) -- every hypothesis ends in a poor local optimum and the reference's two
    builds agree within 1 mm / 1 degree on 21 of 100 only.  The restatement: the same 21."""
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    S, Sn, mx5, mn5 = c1_inputs(hop, golden_dir)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01)
    assert_as_close_as_the_other_build(p, g, slack=4)
    assert (cv == g["converged"]).sum() >= 90


def test_icp_with_the_minimiser_in_exact_arithmetic_vs_the_reference_builds(orc, hop, golden_dir):
    """The same algorithm with every residual in double instead of float (what the GPU's nn_mode 6 evaluates from moment sums):
    without the float run's own rounding noise it lands closer to Eigen's default build than Eigen's -march=native build does."""
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    S, Sn, mx5, mn5 = c2sub_inputs(hop)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01, exact=True)
    mine, native = assert_as_close_as_the_other_build(p, g, slack=3)
    assert mine[0] >= 88 and (it == g["iterations"]).sum() >= 88 and np.array_equal(cv, g["converged"])
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    S, Sn, mx5, mn5 = c1_inputs(hop, golden_dir)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01, exact=True)
    assert_as_close_as_the_other_build(p, g, slack=5)


def test_icp_in_the_moment_form_vs_the_reference_builds(orc, hop, golden_dir):
    """The moment form with integer-exact sums (minimiser 7 = the GPU's nn_mode 7, the form the mirrors run): the 12-bit grid and the
    composed source transform change nothing that can be seen at the scale of the reference's own build-to-build spread -- as close to
    Eigen's default build as the exact-arithmetic form, iteration counts and convergence flags equal on the C2-style subset."""
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    S, Sn, mx5, mn5 = c2sub_inputs(hop)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01, moment=True)
    mine, native = assert_as_close_as_the_other_build(p, g, slack=3)
    assert mine[0] >= 88 and (it == g["iterations"]).sum() >= 88 and np.array_equal(cv, g["converged"])
    pe, ite, cve = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01, exact=True)
    t, r = pose_deltas(p, pe)
    assert ((t < 0.05) & (r < 0.5)).sum() >= len(t) - 3 and (it == ite).sum() >= len(t) - 2   # against the per-point exact form: the grid is invisible
    g = np.load(os.path.join(golden_dir, "icp_lm_c1.npz"))
    S, Sn, mx5, mn5 = c1_inputs(hop, golden_dir)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"], 10, 45.0, 0.01, moment=True)
    assert_as_close_as_the_other_build(p, g, slack=5)
    assert (cv == g["converged"]).sum() >= 88


@pytest.mark.skipif(not os.path.exists("/root/reference"), reason="the reference tree (build container only)")
def test_goldens_are_what_the_reference_build_returns_now(orc, hop, golden_dir, kat):
    """regenerates a slice of the vectors from oracle/_ref/libref_icp.so and compares (guards stale fixtures)"""
    assert orc.ref_icp_available()
    for k in (0, 9):
        P, Q, N = kat[f"lm{k}_P"], kat[f"lm{k}_Q"], kat[f"lm{k}_N"]
        T, x, st = orc.lm_point_to_plane(P, Q, N, ref=True)
        assert biteq(x, kat[f"lm{k}_x"]) and tuple(st) == tuple(int(v) for v in kat[f"lm{k}_stats"])
    g = np.load(os.path.join(golden_dir, "icp_lm_c2sub.npz"))
    S, Sn, mx5, mn5 = c2sub_inputs(hop)
    orc.ref_icp_use(native=False)
    p, it, cv = orc.icp_refine_batch_lm(S, Sn, mx5, mn5, g["poses_in"][:24], 10, 45.0, 0.01, ref=True)
    assert biteq(p, g["poses_out"][:24]) and np.array_equal(it, g["iterations"][:24])
