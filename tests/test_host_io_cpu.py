"""The C++ host's file readers (host/Frame.h: its own 16-bit PNG decoder on zlib, Utils::parsePoseTxt, the calibration chain of
run_real_all.cpp:116-133) against the Python mirror on the same files -- CPU only: nothing here creates a context."""
import os
import subprocess
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    lib = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib")
    if not os.path.exists(os.path.join(lib, "libhop.so")):
        pytest.skip("libhop.so is not built")
    exe = str(tmp_path_factory.mktemp("hostio") / "host_io_check")
    subprocess.run(["g++", "-std=c++17", "-O1", os.path.join(ROOT, "tests", "cpp", "host_io_check.cpp"), "-o", exe, "-L" + lib, "-lhop", "-lz",
                    "-Wl,-rpath," + lib], check=True, capture_output=True)
    return exe


@pytest.mark.parametrize("kind", ["smooth16", "noise16", "gray8"])
def test_cpp_png_pose_and_calibration_readers_equal_the_python_mirror(checker, hop, tmp_path, kind):
    from PIL import Image
    import yaml
    from hop_amd import config as hop_config
    from hop_amd import run_real_all as rr
    rng = np.random.default_rng(3)
    H, W = 97, 131                                       # odd sizes: every PNG filter type sees a partial last group
    if kind == "smooth16":                               # gradients: PIL's encoder picks Sub / Up / Average / Paeth rows
        img = (np.add.outer(np.arange(H) * 37, np.arange(W) * 11) % 60000).astype(np.uint16)
    elif kind == "noise16":
        img = rng.integers(0, 65536, (H, W)).astype(np.uint16)
    else:
        img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    png = str(tmp_path / "depth.png")
    Image.fromarray(img).save(png)
    cfg = hop_config.load_config()
    cfg["cam_K"] = [615.5, 0.0, 322.25, 0.0, 614.75, 241.5, 0.0, 0.0, 1.0]
    q = rng.normal(size=4)
    cfg["cam1_in_leftarm"] = [0.11, -0.03, 0.27] + [float(v) for v in q / np.linalg.norm(q)]
    hp = np.eye(4)
    hp[:3, 3] = [0.01, -0.02, 0.13]
    cfg["handbase_in_palm"] = [float(v) for v in hp.reshape(16)]
    cfg_path = str(tmp_path / "config.yaml")
    with open(cfg_path, "w") as f:
        class _Dumper(yaml.SafeDumper):
            pass
        _Dumper.add_representer(list, lambda d, data: d.represent_sequence("tag:yaml.org,2002:seq", data, flow_style=True))
        yaml.dump(cfg, f, Dumper=_Dumper, default_flow_style=False)

    def pose(seed):
        r = np.random.default_rng(seed)
        A = np.linalg.qr(r.normal(size=(3, 3)))[0]
        T = np.eye(4)
        T[:3, :3] = A * np.sign(np.linalg.det(A))
        T[:3, 3] = r.normal(size=3) * 0.3
        return T
    arm, palm = pose(1), pose(2)
    fmt = lambda T: "\n".join(" ".join(repr(float(v)) for v in row) for row in T) + "\n"
    open(tmp_path / "arm.txt", "w").write(fmt(arm))
    open(tmp_path / "palm.txt", "w").write(fmt(palm))
    r = subprocess.run([checker, cfg_path, png, str(tmp_path / "arm.txt"), str(tmp_path / "palm.txt")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = {ln.split()[0]: ln.split()[1:] for ln in r.stdout.strip().splitlines()}
    flat = img.astype(np.uint64).reshape(-1)
    wsum = int((flat * (np.arange(flat.size, dtype=np.uint64) % np.uint64(9973) + np.uint64(1))).sum())
    assert [int(v) for v in lines["png"]] == [H, W, int(flat.sum()), wsum, int(flat[0]), int(flat[-1])]
    assert np.array_equal(rr.read_depth_png(png).astype(np.uint64).reshape(-1), flat)
    cfg2 = hop_config.load_config(cfg_path)
    K, _, _ = rr.calibration(cfg2)
    assert np.allclose(np.array([float(v) for v in lines["K"]], np.float32), np.asarray(K, np.float32).reshape(9), rtol=0, atol=0)
    hb_py = rr.handbase_in_cam_of(cfg2, rr.parse_pose_txt(str(tmp_path / "arm.txt")), rr.parse_pose_txt(str(tmp_path / "palm.txt")))
    hb_cpp = np.array([float(v) for v in lines["handbase"]], np.float32).reshape(4, 4)
    assert np.abs(hb_cpp - np.asarray(hb_py, np.float32)).max() < 2e-6
    # and against the plain matrix product in double
    q4 = np.array(cfg["cam1_in_leftarm"][3:])
    x, y, z, w = q4 / np.linalg.norm(q4)
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    C = np.eye(4)
    C[:3, :3], C[:3, 3] = R, cfg["cam1_in_leftarm"][:3]
    ref = np.linalg.inv(C) @ (np.linalg.inv(arm) @ palm @ hp)
    assert np.abs(hb_cpp - ref).max() < 1e-5


def test_cpp_config_parser_reads_the_shipped_yaml_like_pyyaml(checker, hop):
    """host/ConfigParser.h (the reference's ConfigParser over yaml-cpp, ConfigParser.cpp) on the shipped config_autodataset.yaml: every scalar
    and every flow list PyYAML finds, under the same dotted key, with the same value"""
    import yaml
    from hop_amd import config as hop_config
    path = os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml")
    r = subprocess.run([checker, path, "dump"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    cpp = dict(ln.split("\t", 1) for ln in r.stdout.splitlines() if "\t" in ln)
    flat = {}

    def walk(prefix, node):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(f"{prefix}.{k}" if prefix else str(k), v)
        else:
            flat[prefix] = node
    walk("", yaml.safe_load(open(path)))
    assert len(flat) > 20
    for k, v in flat.items():
        assert k in cpp, k
        if isinstance(v, list):
            got = [float(x) for x in cpp[k].replace("[", " ").replace("]", " ").replace(",", " ").split()]
            assert got == [float(x) for x in v], k
        elif isinstance(v, bool):
            assert cpp[k].strip().lower() in (("true", "1", "yes") if v else ("false", "0", "no")), k
        elif isinstance(v, (int, float)):
            assert float(cpp[k]) == float(v), k
        else:
            assert cpp[k].strip().strip('"').strip("'") == str(v), k
    assert hop_config.load_config(path)["model_name"] == cpp["model_name"].strip()


def test_cpp_png_reader_rejects_corrupt_headers(checker, tmp_path):
    """A short IHDR chunk or absurd dimensions end in an error message, not in a read past the buffer or a multi-gigabyte allocation
    (Frame.h read_png16 trusted both)."""
    import struct
    import zlib

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xffffffff)
    sig = b"\x89PNG\r\n\x1a\n"
    good_ihdr = struct.pack(">IIBBBBB", 4, 2, 16, 0, 0, 0, 0)
    idat = zlib.compress(b"".join(b"\x00" + bytes(8) for _ in range(2)))
    cases = {"short_ihdr": sig + chunk(b"IHDR", good_ihdr[:9]) + chunk(b"IEND", b""),          # IHDR as the last bytes of the file, too short
             "huge": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0x7fffffff, 0x7fffffff, 16, 0, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b""),
             "zero": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0, 2, 16, 0, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")}
    for name, blob in cases.items():
        png = tmp_path / f"{name}.png"
        png.write_bytes(blob)
        r = subprocess.run([checker, "-", str(png), "-", "-"], capture_output=True, text=True, timeout=60)
        assert r.returncode != 0 and ("IHDR" in r.stderr or "implausible" in r.stderr), (name, r.stdout, r.stderr)
    ok = tmp_path / "ok.png"
    ok.write_bytes(sig + chunk(b"IHDR", good_ihdr) + chunk(b"IDAT", idat) + chunk(b"IEND", b""))
    r = subprocess.run([checker, "-", str(ok), "-", "-"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.split()[:3] == ["png", "2", "4"], r.stdout + r.stderr


def test_cpp_dataset_driver_resumes_and_shards_without_touching_a_device(hop, tmp_path):
    """host/app/run_real_all on a record whose frames all have results: it lists the reference's layout (run_real_all.cpp:72-104), takes
    this rank's share (frame index mod WORLD_SIZE) and, with nothing left to compute, creates no context -- so this runs without a GPU"""
    from hop_amd import run_real_all as rr
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "run_real_all")
    if not os.path.exists(exe):
        pytest.skip("run_real_all is not built")
    base = str(tmp_path / "auto_collect")
    rec = os.path.join(base, "ellipse", "rec_000")
    os.makedirs(rec)
    for k in (0, 1, 2, 5, 12):
        open(os.path.join(rec, f"rgb{k}.png"), "wb").write(b"")          # only the names are read when every frame is done
        os.makedirs(os.path.join(rec, "predict", str(k)))
        open(os.path.join(rec, "predict", str(k), "model2scene.txt"), "w").write("1 0 0 0\n0 1 0 0\n0 0 1 0\n0 0 0 1\n")
    open(os.path.join(rec, "notes.txt"), "w").write("not a frame")
    cfg_path = os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml")
    adir = rr.write_assets_dir(rr.Assets(), str(tmp_path / "assets"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "HOP_FORCE")}
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and "rank 0 of 1: 0 frames written" in r.stdout and "5 resumed" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, env=dict(env, RANK="1", WORLD_SIZE="2", HOP_INFLIGHT="4"))
    assert r.returncode == 0 and "rank 1 of 2: 0 frames written" in r.stdout and "2 resumed" in r.stdout, r.stdout + r.stderr   # frames 1 and 5
    assert not os.path.exists(os.path.join(base, "ellipse", "model2scene_all.txt"))     # no gather unless asked for


def test_cpp_dataset_driver_gathers_the_frame_poses(hop, tmp_path):
    """HOP_GATHER=1 (BASELINE configs[3], "gather of per-frame best pose"): the driver turns its frames' results into the rows of
    hop_frames_allgather, numbers them by the directory listing every rank shares, and rank 0 writes the table.  One rank needs no
    communicator (and, every frame being done, no device): rows, numbering and the written table are what this checks; the collective
    itself is hop_frames_allgather (tests/test_distributed_cpu.py, tests/exchange_child.py)."""
    from hop_amd import run_real_all as rr
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "run_real_all")
    if not os.path.exists(exe):
        pytest.skip("run_real_all is not built")
    base = str(tmp_path / "auto_collect")
    rng = np.random.default_rng(11)
    truth = {}
    for record, frames in (("rec_000", (0, 3, 10, 2)), ("rec_001", (7, 1))):
        rec = os.path.join(base, "ellipse", record)
        os.makedirs(rec)
        for k in frames:
            open(os.path.join(rec, f"rgb{k}.png"), "wb").write(b"")
            os.makedirs(os.path.join(rec, "predict", str(k)))
            T = np.eye(4, dtype=np.float32)
            T[:3, :] = rng.normal(size=(3, 4)).astype(np.float32)
            truth[(record, k)] = T
            with open(os.path.join(rec, "predict", str(k), "model2scene.txt"), "w") as f:
                for row in T:
                    f.write(" ".join(f"{v:.9g}" for v in row) + "\n")
    cfg_path = os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml")
    adir = rr.write_assets_dir(rr.Assets(), str(tmp_path / "assets"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "HOP_FORCE", "MASTER_PORT", "HOP_COMM_ID_FILE")}
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, env=dict(env, HOP_GATHER="1"))
    assert r.returncode == 0 and "poses of 6 frames gathered (6 from this rank, 6 rows per rank)" in r.stdout, r.stdout + r.stderr
    lines = open(os.path.join(base, "ellipse", "model2scene_all.txt")).read().strip().splitlines()
    got = [(ln.split()[0], int(ln.split()[1])) for ln in lines]
    assert got == sorted(truth), "frames in the order of the listing: record, then frame index"
    # the Python runner numbers the frames of its gather from the same listing (run_real_all.py main): rows of the two drivers are interchangeable
    mdir = os.path.join(base, "ellipse")
    py_keys = [(rec, i) for rec in sorted(d for d in os.listdir(mdir) if os.path.isdir(os.path.join(mdir, d))) for i in rr.raw_frame_indices(os.path.join(mdir, rec))]
    assert py_keys == got
    for ln in lines:
        t = ln.split()
        assert np.array_equal(np.array(t[2:], np.float32).reshape(4, 4), truth[(t[0], int(t[1]))])    # %.9g round-trips a float
    # a rank other than 0 waits for rank 0's RCCL id in the shared directory and says so when it never comes
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=60,
                       env=dict(env, HOP_GATHER="1", RANK="1", WORLD_SIZE="2", MASTER_PORT="29999", HOP_COMM_WAIT_S="0.3"))
    assert r.returncode == 3 and "rank 0 did not publish the RCCL id in" in r.stderr and ".hop_comm_id.29999" in r.stderr, r.stdout + r.stderr
    # ... it says so too when the only id in the directory is the one a killed run left behind (older than this launch: ADVICE r03) ...
    stale = os.path.join(mdir, ".hop_comm_id.29999.launch7")
    open(stale, "wb").write(bytes(128))
    old = time.time() - 600
    os.utime(stale, (old, old))
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=60,
                       env=dict(env, HOP_GATHER="1", RANK="1", WORLD_SIZE="2", MASTER_PORT="29999", HOP_RUN_ID="launch7", HOP_COMM_WAIT_S="0.3"))
    assert r.returncode == 3 and "rank 0 did not publish the RCCL id in" in r.stderr and ".hop_comm_id.29999.launch7" in r.stderr, r.stdout + r.stderr
    assert open(os.path.join(mdir, ".hop_comm_id.29999.launch7.abort")).read().startswith("rank 1: rank 0 did not publish"), "a rank that gives up announces it"
    # ... and a rank that finds another rank's announcement stops at once with that rank's reason instead of waiting
    t0 = time.time()
    r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, timeout=60,
                       env=dict(env, HOP_GATHER="1", RANK="1", WORLD_SIZE="2", MASTER_PORT="29999", HOP_RUN_ID="launch7", HOP_COMM_WAIT_S="30"))
    assert r.returncode == 3 and "another rank gave up before the gather: rank 1: rank 0 did not publish" in r.stderr and time.time() - t0 < 10, r.stdout + r.stderr
