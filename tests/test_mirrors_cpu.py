"""Host logic of the mirrors that needs no device: HOP_ICP_NN_MODE parsing and the retry of a refused nn_mode 7."""
import importlib
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.parametrize("value, expect", [(None, 7), ("", 7), ("abc", 7), ("9", 7), ("-1", 7), (" 6 ", 6), ("0", 0), ("5x", 7), ("7", 7)])
def test_hop_icp_nn_mode_is_an_integer_in_range_or_the_default(hop, value, expect, monkeypatch, capsys):
    from hop_amd import api
    if value is None:
        monkeypatch.delenv("HOP_ICP_NN_MODE", raising=False)
    else:
        monkeypatch.setenv("HOP_ICP_NN_MODE", value)
    assert api._icp_nn_mode_reference() == expect
    err = capsys.readouterr().err
    assert ("not an integer in 0..7" in err) == (value not in (None, "") and expect == 7 and value.strip() != "7")


def test_cpp_mirror_reads_the_variable_the_same_way(tmp_path):
    src = tmp_path / "m.cpp"
    src.write_text('#include "%s/icra20-hand-object-pose_amd/host/PoseEstimator.h"\n#include <cstdio>\nint main() { std::printf("%%d\\n", hop::icp_nn_mode_reference()); }\n' % ROOT)
    exe = tmp_path / "m"
    lib = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O0", str(src), "-o", str(exe), "-L" + lib, "-lhop", "-Wl,-rpath," + lib])
    for value, expect in [(None, 7), ("", 7), ("abc", 7), ("9", 7), (" 6 ", 6), ("0", 0), ("5x", 7)]:
        env = {k: v for k, v in os.environ.items() if k != "HOP_ICP_NN_MODE"}
        if value is not None:
            env["HOP_ICP_NN_MODE"] = value
        r = subprocess.run([str(exe)], env=env, capture_output=True, text=True)
        assert r.returncode == 0 and int(r.stdout.strip()) == expect, (value, r.stdout, r.stderr)


def test_a_refused_nn_mode_7_is_retried_once_with_mode_5_and_announced(hop, monkeypatch, capsys):
    from hop_amd import api
    monkeypatch.setattr(api, "ICP_NN_MODE_REFERENCE", 7)
    calls = []

    class Ctx(api.Context):
        def __init__(self):   # no device: only the retry logic is under test
            pass

        def icp_refine(self, max_iter=10, angle_deg=45.0, max_corr_dist=0.01, max_hypotheses=0, nn_mode=0, want_stats=False):
            calls.append(nn_mode)
            if nn_mode == 7:
                e = api.HopError.__new__(api.HopError)
                RuntimeError.__init__(e, "hop_icp_refine: status -5 nn_mode 7 needs the packed cell lists")
                e.status = -5
                raise e
            return "refined"
    assert Ctx().icp_refine_reference(10, 45.0, 0.01, max_hypotheses=100) == "refined"
    assert calls == [7, 5] and "retrying with nn_mode 5" in capsys.readouterr().err

    class Broken(Ctx):
        def icp_refine(self, *a, **k):
            e = api.HopError.__new__(api.HopError)
            RuntimeError.__init__(e, "hop_icp_refine: status -3")
            e.status = -3
            raise e
    with pytest.raises(api.HopError):
        Broken().icp_refine_reference(10, 45.0, 0.01)   # any other failure is not retried
