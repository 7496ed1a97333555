"""Host-side base selection (hop_select.h): the fast exact-guarded path equals the literal restatement of
MatchBase::SelectRandomTriangle / Match4pcsBase::SelectQuadrilateral, and the index draws equal
std::discrete_distribution.  Pure CPU (C++ unit test compiled here)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_selection_fast_path_equals_literal(tmp_path):
    exe = tmp_path / "test_select"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "test_select.cpp"),
                           "-o", str(exe)])
    # both draw implementations: AVX-512 cumulative search (where the CPU has it) and the portable Fenwick / scalar path
    for simd in ("1", "0"):
        env = dict(os.environ, HOP_SELECT_SIMD_DRAW=simd)
        out = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.strip().endswith("OK")


@pytest.fixture(scope="module")
def select_golden_exe(tmp_path_factory):
    exe = tmp_path_factory.mktemp("selgold") / "select_golden"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", os.path.join(ROOT, "tests", "cpp", "select_golden.cpp"), "-o", str(exe)])
    return str(exe)


@pytest.mark.parametrize("case", ["case1", "case2", "depth7", "obj_cuboid", "obj_cylinder", "obj_tless3", "obj_mustard"])
def test_product_host_selection_against_the_reference_build(select_golden_exe, golden_dir, tmp_path, case):
    """The PRODUCT's host code (csrc/hop_select.h: MatchBase::init sampling / centring, the PPF key membership of matchBase.hpp:31-68,
    SelectRandomTriangle / SelectQuadrilateral, matchBase.hpp:111-212, match4pcsBase.hpp:50-189) on the inputs of the reference-built
    golden vectors, without a device: every base the reference build traced -- it records one entry per successful generateCongruents(),
    oracle/ref_driver.cpp -- appears in the product's 30 trials in the same order with the same four ids and bit-equal invariants; the
    trials in between are those whose congruent set came out empty.  Both the fast path and the literal restatement."""
    g = np.load(os.path.join(golden_dir, f"s4pcs_{case}.npz"))
    assert int(g["opts"][2]) == 1
    dump = str(tmp_path / "in.bin")
    with open(dump, "wb") as f:
        np.array([len(g["P_xyz"]), len(g["Q_xyz"]), len(g["keys"])], np.int32).tofile(f)
        for a in (g["P_xyz"], g["P_nrm"]):
            np.ascontiguousarray(a.T.astype(np.float32)).tofile(f)
        g["P_conf"].astype(np.float32).tofile(f)
        for a in (g["Q_xyz"], g["Q_nrm"]):
            np.ascontiguousarray(a.T.astype(np.float32)).tofile(f)
        g["keys"].astype(np.int32).tofile(f)
    outs = []
    for fast in ("1", "0"):
        r = subprocess.run([select_golden_exe, dump, str(int(g["opts"][0])), fast], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("base ")]
        ids = np.array([[int(v) for v in x[:4]] for x in rows], np.int32).reshape(-1, 4)
        inv = np.array([[int(v, 16) for v in x[4:]] for x in rows], np.uint32).reshape(-1, 2).view(np.float32)
        outs.append((ids, inv))
        # sampling, centring, diameter (rows B0 / B2): bit-equal to the reference build
        st = [ln.split()[1:] for ln in r.stdout.splitlines() if ln.startswith("state ")][0]
        f32 = lambda xs: np.array([int(v, 16) for v in xs], np.uint32).view(np.float32)
        assert int(st[0]) == len(g["Qs"])
        assert np.array_equal(f32(st[1:4]), g["cP"]) and np.array_equal(f32(st[4:7]), g["cQ"]) and f32(st[7:8])[0] == g["diameter"]
        q = np.array([[int(v, 16) for v in ln.split()[1:]] for ln in r.stdout.splitlines() if ln.startswith("q ")], np.uint32).view(np.float32)
        assert np.array_equal(q[:, :3], g["Qs"]) and np.array_equal(q[:, 3:], g["Qs_nrm"])
        k, skipped = 0, 0
        for b, iv in zip(g["base_ids"], g["base_inv"]):
            while k < len(ids) and not (np.array_equal(ids[k], b) and np.array_equal(inv[k], iv)):
                k, skipped = k + 1, skipped + 1
            assert k < len(ids), f"base {b} of the reference build is not among the product's selections (fast={fast})"
            k += 1
        assert np.array_equal(ids[0], g["base_ids"][0]) or skipped > 0
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])     # fast path == literal restatement
