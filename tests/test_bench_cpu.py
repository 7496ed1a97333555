"""bench.py's host logic that needs no device: the crash / hang guard's fallback order and report."""
import json
import os
import subprocess
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _args(nn_mode):
    return types.SimpleNamespace(nn_mode=nn_mode, lcp_mode=3, verify_mode=2)


def test_preflight_keeps_the_requested_mode_when_its_child_survives(monkeypatch):
    import bench
    calls = []

    def fake_run(cmd, **kw):
        calls.append(cmd)
        return types.SimpleNamespace(returncode=0, stdout="noise\n" + json.dumps({"ok": True, "hypotheses": 5}) + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    mode, rep = bench.preflight(_args(7), 0)
    assert mode == 7 and rep["timed_nn_mode"] == 7 and len(calls) == 1 and rep["tries"][0]["ok"]
    assert "--preflight-child" in calls[0] and calls[0][calls[0].index("--nn-mode") + 1] == "7"


def test_preflight_falls_back_through_the_hardware_proven_modes_and_says_so(monkeypatch):
    import bench
    seen = []

    def fake_run(cmd, **kw):
        m = int(cmd[cmd.index("--nn-mode") + 1])
        seen.append(m)
        if m == 7:
            return types.SimpleNamespace(returncode=-11, stdout="", stderr="Memory access fault by GPU node-1")   # the child died
        if m == 6:
            raise subprocess.TimeoutExpired(cmd, 1)                                                              # the child hung
        return types.SimpleNamespace(returncode=0, stdout=json.dumps({"ok": True}) + "\n", stderr="")
    monkeypatch.setattr(subprocess, "run", fake_run)
    mode, rep = bench.preflight(_args(7), 0)
    assert seen == [7, 6, 4] and mode == 4 and rep["requested_nn_mode"] == 7 and rep["timed_nn_mode"] == 4
    assert [t["ok"] for t in rep["tries"]] == [False, False, True] and "fault" in rep["tries"][0]["stderr_tail"] and "timeout" in rep["tries"][1]["error"]


def test_preflight_of_a_gauss_newton_mode_has_no_fallback_and_a_dead_child_leaves_the_mode_alone(monkeypatch):
    import bench
    monkeypatch.setattr(subprocess, "run", lambda cmd, **kw: types.SimpleNamespace(returncode=1, stdout="", stderr="boom"))
    mode, rep = bench.preflight(_args(4), 0)
    assert mode == 4 and len(rep["tries"]) == 1 and not rep["tries"][0]["ok"]   # the parent then runs mode 4 and fails visibly itself
