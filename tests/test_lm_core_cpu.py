"""nn_mode 7 (the moment form of the reference's ICP minimiser with integer-exact sums), the part that needs no GPU.

  * the oracle's moment form (oracle/hop_oracle.cpp "The moment form", minimiser 7) is PINNED to Eigen's own run like the other statements of
    the minimiser: on the correspondence sets of tests/golden/icp_lm_kat.npz (made by oracle/ref_icp_driver.cpp from the reference's vendored
    Eigen) it reaches Eigen's minimum -- same cost within the minimiser's ftol, parameters as close as the float restatement's;
  * the integer moment sums do not depend on the order of the correspondences (what a GPU's lanes / wavefronts / workgroups change);
  * the PRODUCT's minimiser text -- csrc/hop_lm_core.h, the header k_icp_lm7_solve runs -- compiled by g++ (tests/cpp/lm_core_host.cpp) returns
    the oracle's parameters, status and evaluation count BIT FOR BIT on hundreds of moment matrices, including rank-deficient and
    zero-residual ones: two independent texts, same doubles.  The kernel around it (loads, the 64-bit block sums) is checked on the GPU
    (tests/test_gpu_icp_lm.py).
"""
import os
import struct
import subprocess

import numpy as np
import pytest

from test_icp_lm_oracle import _fnorm, biteq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def kat(golden_dir):
    return np.load(os.path.join(golden_dir, "icp_lm_kat.npz"))


@pytest.fixture(scope="module")
def core_host(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("lmcore") / "lm_core_host")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-Wno-unknown-pragmas", os.path.join(ROOT, "tests", "cpp", "lm_core_host.cpp"), "-o", exe],
                   check=True, capture_output=True)
    return exe


def _moments(orc, P, Q, N, bits=12, ctr=None):
    ctr = Q.mean(0).astype(np.float32) if ctr is None else ctr
    radius = float(np.sqrt(((P.astype(np.float64) - ctr) ** 2).sum(1).max()))
    d2 = ((P - Q) ** 2).sum(1).astype(np.float32)
    gate = float(np.sqrt(d2.max()) * 1.001 + 1e-6)
    Mi, dq, Md, sc = orc.mom_accumulate(P, Q, N, d2, ctr, radius, gate, bits)
    return Mi, Md, ctr, sc


def _sets(kat):
    return [(kat[f"lm{k}_P"], kat[f"lm{k}_Q"], kat[f"lm{k}_N"], kat[f"lm{k}_x"]) for k in range(int(kat["n_lm_sets"]))]


def test_moment_form_reaches_eigens_minimum(orc, kat):
    for k, (P, Q, N, xg) in enumerate(_sets(kat)):
        _, Md, ctr, _ = _moments(orc, P, Q, N)
        T, x, st = orc.lm_point_to_plane_moments(Md, ctr)
        assert st[0] in (1, 2, 3), (k, st)
        assert biteq(T, orc.lm_warp6(x))
        c_mine, c_gold, c_start = _fnorm(orc, P, Q, N, x), _fnorm(orc, P, Q, N, xg), _fnorm(orc, P, Q, N, np.zeros(6, np.float32))
        assert c_mine < c_start
        assert abs(c_mine - c_gold) / c_gold < 5e-3, (k, c_mine, c_gold)    # the tolerance of test_restated_minimiser_reaches_eigens_minimum
        assert np.abs(x - xg).max() < (5e-4 if k >= 8 else 3e-2), (k, np.abs(x - xg).max())


def test_moment_matrix_is_the_quadratic_form_of_the_residuals(orc, kat):
    """w(x)^T M w(x) = sum f_i(x)^2 up to the grid: the algebra the form rests on, checked against the float residuals Eigen evaluates"""
    P, Q, N, xg = _sets(kat)[9]
    _, Md, ctr, _ = _moments(orc, P, Q, N, bits=20)
    for x in (np.zeros(6, np.float32), xg, (xg * 0.5).astype(np.float32)):
        f, _ = orc.lm_residuals_jacobian(P, Q, N, x)
        T = orc.lm_warp6(x).astype(np.float64)
        w = np.concatenate([(T[:3, :3] - np.eye(3)).reshape(9), T[:3, 3] + (T[:3, :3] - np.eye(3)) @ ctr.astype(np.float64), [1.0]])
        assert abs(w @ Md @ w - float((f.astype(np.float64) ** 2).sum())) <= 2e-4 * float((f.astype(np.float64) ** 2).sum()) + 1e-12


def test_integer_moments_do_not_depend_on_the_order(orc, kat):
    rng = np.random.default_rng(5)
    for P, Q, N, _ in _sets(kat)[:6]:
        Mi, Md, ctr, sc = _moments(orc, P, Q, N)
        for _ in range(3):
            o = rng.permutation(len(P))
            Mi2, Md2, _, _ = _moments(orc, P[o], Q[o], N[o], ctr=ctr)
            assert np.array_equal(Mi, Mi2) and np.array_equal(Md, Md2)
        assert np.abs(Mi).max() < 2 ** 53 and np.array_equal(Mi, Mi.T)


def _cases(orc, kat):
    """moment matrices of real correspondence sets, of random subsets of them, and the degenerate shapes a frame can produce"""
    rng = np.random.default_rng(17)
    out = []
    sets = _sets(kat)
    for P, Q, N, _ in sets:
        out.append(_moments(orc, P, Q, N)[1:3])
        for _ in range(12):
            m = int(rng.integers(4, len(P)))
            o = rng.choice(len(P), m, replace=False)
            out.append(_moments(orc, P[o], Q[o], N[o])[1:3])
    P, Q, N, _ = sets[0]
    n1 = np.tile(np.array([[0.0, 0.0, 1.0]], np.float32), (len(P), 1))
    out.append(_moments(orc, P, Q, n1)[1:3])                                   # every normal the same: three free parameters
    out.append(_moments(orc, P, P.copy(), N)[1:3])                            # zero residuals: the gradient test stops the run
    out.append(_moments(orc, P[:4], Q[:4], N[:4])[1:3])                       # four correspondences: the fewest PCL accepts
    flat = P.copy()
    flat[:, 2] = flat[:, 2].mean()
    out.append(_moments(orc, flat, Q, n1)[1:3])                               # a plane against a plane
    out.append(_moments(orc, P + np.float32(0.004), Q, N)[1:3])               # a large offset: the trust region is active (par > 0)
    # the nine ICP iterations of a poor C1 hypothesis; the last one sends a trial step outside the quaternion's unit ball (NaN residuals): Eigen
    # refuses such a step because comparisons with NaN are false -- round 3's fmax(ff, 0) had turned it into "a perfect step" (see hop_lm_core.h)
    g = np.load(os.path.join(ROOT, "tests", "golden", "lm_moment_c1_hyp78.npz"))
    out += [(M, c) for M, c in zip(g["M"], g["c"])]
    return out


def test_product_minimiser_text_equals_the_oracle_bit_for_bit(orc, kat, core_host, tmp_path):
    cases = _cases(orc, kat)
    path = tmp_path / "cases.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(cases)))
        for Md, ctr in cases:
            f.write(np.ascontiguousarray(Md, np.float64).tobytes())
            f.write(np.asarray(ctr, np.float64).tobytes())
    r = subprocess.run([core_host, str(path)], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    counts = [int(v) for v in lines.pop().split()[1:]]
    assert len(lines) == len(cases)
    assert min(counts) > 0, ("every lmpar2 path ran: general / pivoted, register with par = 0, register with the secular iteration", counts)
    n_par = 0
    statuses = set()
    for k, ((Md, ctr), ln) in enumerate(zip(cases, lines)):
        t = ln.split()
        x_host = np.array([int(v, 16) for v in t[:6]], np.uint32).view(np.float32)
        _, x, st = orc.lm_point_to_plane_moments(Md, ctr)
        assert np.array_equal(x_host.view(np.uint32), x.view(np.uint32)), (k, x_host, x)
        assert tuple(int(v) for v in t[6:9]) == st, (k, t[6:9], st)
        statuses.add(st[0])
        n_par += st[1] > 8
        assert np.isfinite(x).all(), k
    assert len(cases) > 150 and n_par > 100 and len(statuses) >= 2, (len(cases), n_par, statuses)


def test_fast_arithmetic_machine_refuses_a_nan_step_too(orc, core_host, tmp_path):
    """nn_mode 6's branch of lm_advance mapped a NaN sum of squares to fnorm1 = 0 ("a perfect step") after round 4 had fixed the
    same defect in lm6_eval.  The nine moment matrices of C1 hypothesis 78 (the last sends a trial step outside the quaternion's unit ball)
    through LmDev6 on the host: the step must be refused as Eigen refuses it -- finite parameters, the status and evaluation count of the
    IEEE machine (258 evaluations, status 1; the defect stopped at 16 with status 4 and NaN parameters)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "lm_moment_c1_hyp78.npz"))
    path = tmp_path / "hyp78.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(g["M"])))
        for Md, ctr in zip(g["M"], g["c"]):
            f.write(np.ascontiguousarray(Md, np.float64).tobytes())
            f.write(np.asarray(ctr, np.float64).tobytes())
    out = {}
    for mode in ("7", "6"):
        r = subprocess.run([core_host, str(path)] + (["6"] if mode == "6" else []), capture_output=True, text=True, check=True)
        out[mode] = [ln.split() for ln in r.stdout.strip().splitlines()[:-1]]
    for k, (t7, t6) in enumerate(zip(out["7"], out["6"])):
        x6 = np.array([int(v, 16) for v in t6[:6]], np.uint32).view(np.float32)
        x7 = np.array([int(v, 16) for v in t7[:6]], np.uint32).view(np.float32)
        assert np.isfinite(x6).all(), (k, x6)
        assert t6[6:9] == t7[6:9], (k, t6[6:9], t7[6:9])       # same status, evaluations and iterations as the IEEE machine
        assert np.abs(x6 - x7).max() <= 1e-6, (k, x6, x7)      # (on the host the two differ by the order of a few double operations)


def test_product_minimiser_text_equals_the_oracle_on_random_problems(orc, core_host, tmp_path):
    """1 500 random point-to-plane problems -- surface patches of random extent and curvature, increments from 0.01 to 30 degrees / 0.1 to 40 mm
    (the large ones drive trial steps through the trust-region logic and, now and then, outside the quaternion's unit ball), noise from 0 to
    3 mm, 4 to 400 correspondences: parameters, status and evaluation count of the product's text equal the oracle's bit for bit"""
    rng = np.random.default_rng(99)
    cases = []
    for k in range(1500):
        m = int(rng.integers(4, 400))
        ext = rng.uniform(0.005, 0.08, 3)
        P = (rng.normal(size=(m, 3)) * ext).astype(np.float32)
        if k % 7 == 0:
            P[:, 2] = (0.5 * rng.uniform(-20, 20) * (P[:, 0] ** 2 + P[:, 1] ** 2)).astype(np.float32)      # a curved sheet
        N = P / np.maximum(ext ** 2, 1e-9) + rng.normal(size=(m, 3)) * 0.05
        N = (N / np.linalg.norm(N, axis=1, keepdims=True)).astype(np.float32)
        ang = np.radians(10 ** rng.uniform(-2, np.log10(30.0)))
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        t = rng.normal(size=3) * 10 ** rng.uniform(-4, np.log10(0.04))
        ctr0 = rng.normal(size=3) * 0.3
        Q = ((P.astype(np.float64) @ R.T + t) + rng.normal(size=(m, 3)) * rng.uniform(0, 0.003) * (k % 3 > 0) + ctr0).astype(np.float32)
        Pm = (P + ctr0).astype(np.float32)
        cases.append(_moments(orc, Pm, Q, N)[1:3])
    path = tmp_path / "fuzz.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<i", len(cases)))
        for Md, ctr in cases:
            f.write(np.ascontiguousarray(Md, np.float64).tobytes())
            f.write(np.asarray(ctr, np.float64).tobytes())
    r = subprocess.run([core_host, str(path)], capture_output=True, text=True, check=True)
    lines = r.stdout.strip().splitlines()
    counts = [int(v) for v in lines.pop().split()[1:]]
    assert len(lines) == len(cases) and min(counts) > 0, counts
    statuses = {}
    for k, ((Md, ctr), ln) in enumerate(zip(cases, lines)):
        t = ln.split()
        x_host = np.array([int(v, 16) for v in t[:6]], np.uint32).view(np.float32)
        _, x, st = orc.lm_point_to_plane_moments(Md, ctr)
        assert np.array_equal(x_host.view(np.uint32), x.view(np.uint32)) and tuple(int(v) for v in t[6:9]) == st, (k, x_host, x, t[6:9], st)
        statuses[st[0]] = statuses.get(st[0], 0) + 1
    assert len(statuses) >= 3, statuses
