"""CPU-only checks of the boundary: the C-ABI library builds for gfx950, loads, and exports exactly what
include/hop.h declares; host-side pure functions; the YAML surface.  No kernel is launched here."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def api(hop):
    from hop_amd import api as _api
    _api.build_library()
    return _api


def _header_functions():
    text = open(os.path.join(ROOT, "include", "hop.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(hop_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(api):
    L = api.lib()
    declared = _header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/hop.h but not exported by libhop.so"
    # the binding covers the same set (no silently unbound entry point)
    assert sorted(api.SIGNATURES) == declared
    import re
    ver = int(re.search(r"#define HOP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "hop.h")).read()).group(1))
    assert L.hop_abi_version() == ver == api.HOP_ABI_VERSION >= 2   # bumped when entry points or struct layouts change (round 3 added five)


def test_every_entry_point_is_bound_in_integration_md_and_cites_the_reference():
    """The drop-in boundary is documented entry point by entry point: INTEGRATION.md names every function of include/hop.h, and the
    header cites reference files (file:line) throughout."""
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [n for n in _header_functions() if not re.search(r"\b" + n + r"\b", integ)]
    assert not missing, f"not named in INTEGRATION.md: {missing}"
    header = open(os.path.join(ROOT, "include", "hop.h")).read()
    cites = re.findall(r"[A-Za-z0-9_]+\.(?:cpp|hpp|h|yaml|py):\d+", header)
    assert len(cites) >= 60, len(cites)


def test_library_contains_gfx950_code_object(api):
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", api.LIB_PATH], capture_output=True, text=True)
    blob = open(api.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert b"k_verify_brute" in blob and b"k_icp_nn" in blob and b"k_lcp_forward" in blob
    assert b"k_phys_fingers" in blob and b"k_sdf_query" in blob and b"k_sor_mean" in blob
    assert out.returncode == 0


def test_struct_layouts_match_header(api, tmp_path):
    """sizeof() of every ABI struct as plain C (gcc) sees include/hop.h == the ctypes mirror."""
    names = {"hop_s4pcs_opts": api.S4pcsOpts, "hop_s4pcs_stats": api.S4pcsStats, "hop_icp_opts": api.IcpOpts,
             "hop_lcp_opts": api.LcpOpts, "hop_finger_args": api.FingerArgs, "hop_pso_settings": api.PsoSettings,
             "hop_timing": api.Timing, "hop_physics_args": api.PhysicsArgs, "hop_hand_link": api.HandLink}
    src = tmp_path / "sz.c"
    body = "".join(f'  printf("{n} %zu\\n", sizeof({n}));\n' for n in names)
    src.write_text('#include <stdio.h>\n#include "hop.h"\nint main(void) {\n' + body + "  return 0;\n}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    for line in out.strip().splitlines():
        n, sz = line.split()
        assert C.sizeof(names[n]) == int(sz), n


def test_no_device_is_an_error_not_a_fallback(api):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(api.HopError) as e:
        api.Context(0)
    assert e.value.status == -2


def test_strerror(api):
    L = api.lib()
    assert L.hop_strerror(0) == b"ok"
    assert b"no CPU path" in L.hop_strerror(-2)


def test_topk_merge_is_hypocompare_order(api):
    rng = np.random.default_rng(0)
    k = 16
    tables = np.zeros((3, k, api.TOPK_ROW_FLOATS), np.float32)
    allrows = []
    for t in range(3):
        sc = np.sort(rng.integers(0, 6, k).astype(np.float32))[::-1] / 6.0
        for r in range(k):
            tables[t, r, 0] = sc[r]
            tables[t, r, 1:2] = np.array([t * 1000 + r], np.int32).view(np.float32)
            tables[t, r, 2:] = rng.normal(size=16)
            allrows.append((sc[r], t * 1000 + r))
    tables[2, k - 3:, 1] = np.array([-1], np.int32).view(np.float32)[0]  # padding rows
    allrows = [r for r in allrows if not (r[1] >= 2000 + k - 3)]
    out, n = api.topk_merge(tables, k)
    exp = sorted(allrows, key=lambda r: (-r[0], r[1]))[:k]
    pose, score, ids = api.rows_to_hypos(out)
    assert n == k
    assert [(float(s), int(i)) for s, i in zip(score, ids)] == [(float(a), b) for a, b in exp]


def test_topk_merge_orders_by_the_canonical_key(api):
    """the device merge sorts by a bit-pattern key, the host by float comparison -- they disagreed on -0 vs +0 and on NaN.  Both
    now order by hop::score_order_key (csrc/hop_math.h): -0 counts as +0 (ties by id), a NaN score sorts below every number; this is
    the host side (the device side: tests/test_gpu_pipeline.py, same rows)."""
    k = 6
    scores = [0.0, -0.0, float("nan"), 1.5, -2.0, float("-inf")]
    ids = [5, 4, 1, 9, 2, 3]
    t = np.zeros((1, k, api.TOPK_ROW_FLOATS), np.float32)
    for r, (sc, i) in enumerate(zip(scores, ids)):
        t[0, r, 0] = sc
        t[0, r, 1:2] = np.array([i], np.int32).view(np.float32)
    out, n = api.topk_merge(t, k)
    assert n == k
    assert out.reshape(k, -1)[:, 1].copy().view(np.int32).tolist() == [9, 4, 5, 2, 3, 1]   # 1.5 | -0 (id 4) before +0 (id 5) | -2 | -inf | NaN last


def test_cluster_poses_host_matches_oracle(api, orc, hop):
    synth = hop.synth
    sc = synth.make_scene(200, seed=3)
    poses = synth.replay_poses(sc.gt_pose, 300, seed=2)
    rng = np.random.default_rng(1)
    scores = (rng.integers(0, 30, len(poses)) / 30).astype(np.float32)
    ids = rng.permutation(len(poses)).astype(np.int32)
    for angle, dist, sym in ((30.0, 0.015, [180, 180, 180]), (5.0, 0.003, [180, 180, 0]), (10.0, 0.01, [360, 360, 360])):
        a = api.cluster_poses_host(poses, scores, ids, angle, dist, sym)
        b = orc.cluster_poses(poses, scores, ids, angle, dist, sym)
        assert np.array_equal(a, b)
        assert 1 <= len(a) <= len(poses)


def test_finger_property_matches_oracle(api, orc, hop):
    hand = hop.synth.t42_hand()
    for name in ("finger_1_1", "finger_2_2"):
        xyz = hand.clouds[name][0]
        fp = api.finger_property(xyz, 10)
        X = orc.soa(xyz)
        mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
        st = np.zeros(1, np.float32)
        hist = np.zeros((6, 10), np.float32)
        orc.lib().orc_finger_property(orc.F(X), X.shape[1], 10, orc.F(mn), orc.F(mx), orc.F(st), orc.F(hist))
        assert np.array_equal(fp["min"], mn) and np.array_equal(fp["max"], mx)
        assert fp["stride_z"] == st[0] and np.array_equal(fp["hist"], hist)


def test_config_surface_python_and_cpp_agree(api, hop):
    from hop_amd import config
    cfg = config.load_config()
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "main_realdata_auto")
    out = subprocess.run([exe, "--dump-config", config.DEFAULT_CONFIG], capture_output=True, text=True, check=True).stdout
    kv = dict(line.split("=", 1) for line in out.strip().splitlines())
    assert float(kv["super4pcs_delta"]) == cfg["super4pcs_delta"] == 0.003
    assert int(kv["hand_match.pso.n_pop"]) == cfg["hand_match"]["pso"]["n_pop"] == 15
    assert float(kv["object_symmetry.cuboid.z"]) == cfg["object_symmetry"]["cuboid"]["z"] == 90
    assert float(kv["lcp.dist"]) == cfg["lcp"]["dist"] and float(kv["icp_angle_thres"]) == cfg["icp_angle_thres"]
    assert kv["hand_match.check_normal"] == "true" and cfg["hand_match"]["check_normal"] is True
    with pytest.raises(KeyError):
        import tempfile
        import yaml
        bad = dict(cfg)
        del bad["lcp"]
        with tempfile.NamedTemporaryFile("w", suffix=".yaml", delete=False) as f:
            yaml.safe_dump(bad, f)
        config.load_config(f.name)


def test_reference_config_as_shipped_is_accepted(hop):
    """Only in the build container (the reference tree is not present on the GPU box)."""
    ref = "/root/reference/config_autodataset.yaml"
    if not os.path.exists(ref):
        pytest.skip("reference tree absent")
    from hop_amd import config
    cfg = config.load_config(ref)
    ours = config.load_config()
    for k in config.REQUIRED:
        if k == "object_symmetry":
            for m in ("ellipse", "cylinder", "cuboid", "tless3"):
                assert cfg[k][m] == ours[k][m]
        else:
            assert cfg[k] == ours[k], k
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "main_realdata_auto")
    out = subprocess.run([exe, "--dump-config", ref], capture_output=True, text=True, check=True).stdout
    kv = dict(line.split("=", 1) for line in out.strip().splitlines())
    assert kv["model_name"] == "ellipse" and float(kv["super4pcs_dispersion"]) == 0.5
    assert len(kv["cam_K"].replace("[", " ").replace("]", " ").replace(",", " ").split()) == 9
    assert len(kv["handbase_in_palm"].replace("[", " ").replace("]", " ").replace(",", " ").split()) == 16


def test_moment_grid_width_is_one_number(orc):
    """the grid of nn_mode 7's moment form is stated in three places -- the product (csrc/hop_device.h ICP_MOM_BITS), the oracle's
    default (g_mom_bits) and orc.py's helper default; a test that calls orc.mom_accumulate with its default must compare the kernel's grid."""
    dev = open(os.path.join(ROOT, "icra20-hand-object-pose_amd", "csrc", "hop_device.h")).read()
    bits = int(re.search(r"constexpr int ICP_MOM_BITS = (\d+);", dev).group(1))
    ora = open(os.path.join(ROOT, "oracle", "hop_oracle.cpp")).read()
    assert int(re.search(r"int g_mom_bits = (\d+);", ora).group(1)) == bits
    assert int(re.search(r"int mom_bits = (\d+);", ora).group(1)) == bits
    assert orc.MOM_BITS == bits
    import inspect
    assert inspect.signature(orc.mom_accumulate).parameters["bits"].default == bits
