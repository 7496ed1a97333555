"""Host-side base selection (hop_select.h): the fast exact-guarded path equals the literal restatement of
MatchBase::SelectRandomTriangle / Match4pcsBase::SelectQuadrilateral, and the index draws equal
std::discrete_distribution.  Pure CPU (C++ unit test compiled here)."""
import os
import subprocess

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
