#!/usr/bin/env python
"""bench.py -- one JSON line for the driver (see the contract in the task statement).

Workload (BASELINE.json configs[1], "C2"): ellipse object, one MI355X per rank, 2048 Super4PCS base trials +
200-particle hand-state search, ~20k-point scene / ~5k-point model, all synthetic and seeded
(hop_amd.synth).  One step = one frame of the hot path:

    4 x Hand::matchOneComponentPSO (200(+1) particles x (1+3) evaluations each)
    PoseEstimator::runSuper4pcs with n_trials = 2048           -> H_gen verified hypotheses (resident)
    keep the H = 10240 best by Verify-LCP                       (top-k, HypoCompare order)
    PoseEstimator::refineByICP on all H                         (<=10 point-to-plane iterations each)
    PoseEstimator::selectBest = Utils::computeLCP on all H      -> best pose
    [N > 1] all-gather of each rank's top-128 table + merge     (RCCL through torch.distributed)

value = (hypotheses that went through generation+Verify, ICP and computeLCP, summed over ranks) / (max-over-ranks
wall time).  Ranks draw different base sequences (random_seed + rank): hypothesis-parallel weak scaling.
Inputs are resident in HBM before the timed region; the timed region ends with the small result read-back.
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# the host driver of the GPU boxes supports dmabuf IPC only: without this RCCL's cross-process buffer sharing fails with
# `hipIpcGetMemHandle: invalid argument`.  Exported by the image already; kept here for any launcher that builds its own environment
# (must be in place before the HSA runtime starts, i.e. before torch / libhop.so are loaded).
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=64)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--scene", type=int, default=20000)
    ap.add_argument("--model", type=int, default=5000)
    ap.add_argument("--bases", type=int, default=2048)
    ap.add_argument("--hyps", type=int, default=10240)
    ap.add_argument("--particles", type=int, default=200)
    ap.add_argument("--hand-scene", type=int, default=20000)
    ap.add_argument("--verify-mode", type=int, default=2, help="0 brute-force LDS scan, 1 voxel grid, 2 EXIST-mode cell lists (identical counts)")
    ap.add_argument("--nn-mode", type=int, default=7, help="ICP: 0 brute force, 1 voxel grids, 2 NN cell lists, 3 packed NN cell lists "
                    "with search and accumulation in one kernel (identical correspondences in modes 0-3), 4 = 3 with the increments "
                    "composed into one transform per iteration (poses equal to ~1e-6, tests/test_gpu_fullsize.py) -- modes 0-4: one Gauss-Newton "
                    "step per iteration; 5 / 6 / 7: the reference's Levenberg-Marquardt minimiser (float-faithful passes / float moment sums / "
                    "integer-exact moment sums + IEEE solve = the oracle's bits: the default and what the mirrors run)")
    ap.add_argument("--lcp-mode", type=int, default=3, help="computeLCP: 0 brute force, 1 voxel grids, 2 NN cell lists with the reference's ordered "
                    "float sum (bit-equal scores in modes 0-2), 3 NN cell lists with in-wave partial sums (scores within 1e-4 relative)")
    ap.add_argument("--pso-sum-mode", type=int, default=1, help="outer-side penalty of objFuncPSO: 0 added in scene order (the reference's float "
                    "sum, bit-equal), 1 block reduction (equal to ~1e-6 relative)")
    ap.add_argument("--inflight", type=int, default=8, help="frames in flight per GPU (one context each): the host base selection of one "
                    "frame overlaps the device work of the others; 1 = strictly one frame at a time")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, BASELINE configs[1] 'C2' per GPU: every rank runs its own 2048 base trials) or strong (configs[4] "
                         "'C5': a FIXED replay set of --strong-hyps hypotheses, seed 13, on a 50k-point scene, split contiguously over the ranks)")
    ap.add_argument("--strong-hyps", type=int, default=65536)
    ap.add_argument("--strong-scene", type=int, default=50000)
    ap.add_argument("--no-serial-frame", action="store_true", help="skip the extra undisturbed frame used for per-kernel timing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-alt-modes", action="store_true", help="skip the short runs of the other ICP / sum configurations after the timed region")
    ap.add_argument("--no-next-rows", action="store_true", help="skip the separate measurement of the physics rejection (SURVEY 8f N1)")
    ap.add_argument("--physics-hyps", type=int, default=2048)
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--parity-sample", type=int, default=64, help="hypotheses of one more frame (same seed, same stages as a timed step) whose refined poses the "
                    "oracle recomputes after the timed region: the bench line's parity gate (BASELINE.md 3.6); 0 = skip")
    ap.add_argument("--no-preflight", action="store_true", help="skip the crash / hang guard that runs one small frame in a child process first")
    ap.add_argument("--preflight-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--preflight-device", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args()


def preflight_child(args):
    """One SMALL frame of the hot path (generate -> keep -> refineByICP in the requested nn_mode -> computeLCP) on the device, in a process
    of its own: a kernel that faults or hangs on first contact with a part takes this child down, not the bench.  No oracle here -- parity
    is judged by parity_sample after the timed region; this only says "it runs, the numbers are finite, and the library's own device
    self-checks passed"."""
    import hop_loader
    hop = hop_loader.load()
    from hop_amd import api
    synth = hop.synth
    ctx = api.Context(args.preflight_device)
    chk = ctx.selfcheck(force=True)
    sc = synth.make_scene(2000, seed=7)
    mx5, mn5 = synth.ellipsoid_model_spacing(0.005)
    mx1, mn1 = synth.ellipsoid_model(3000)
    ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    ctx.set_model(api.HOP_MODEL_5MM, mx5, mn5)
    ctx.set_model(api.HOP_MODEL_1MM, mx1, mn1)
    ctx.set_ppf_keys(synth.ppf_key_table())
    o = ctx.default_s4pcs_opts(sample_size=100, success_quadrilaterals=64, max_time_seconds=0, n_trials=64, random_seed=5489, verify_mode=args.verify_mode)
    ctx.s4pcs_generate(o, download=False)
    ctx.hypos_keep_topk(512)
    n = ctx.hypos_count()
    it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=args.nn_mode, want_stats=True)
    best, score, idx = ctx.lcp_select_best(0.001, 10.0, args.lcp_mode)
    p, s_, _ = ctx.hypos_download()
    chk2 = ctx.selfcheck(force=False)
    print(json.dumps({"ok": bool(n > 0 and np.isfinite(p).all() and np.isfinite(s_).all() and np.isfinite(score)), "hypotheses": int(n),
                      "icp_iterations_mean": float(np.mean(it)) if n else 0.0, "best_lcp_score": float(score),
                      "selfcheck": {"packed_ranking_q_rank": chk["qrank"], "matrix_core_read_out": chk["mfma"]}, "icp_engine": chk2["icp_engine"]}), flush=True)
    ctx.close()


def preflight(args, device):
    """Runs preflight_child for args.nn_mode; if that child dies (fault), hangs (timeout) or reports garbage, tries the ICP forms that HAVE
    run on this hardware before (nn_mode 6: round 3, nn_mode 4: round 2) in turn.  Returns (nn_mode to time, report for the JSON line).  A
    fallback is never silent: the line carries the report, `config.nn_mode` is the mode that was timed, and `dtype` / `icp_minimiser` follow it."""
    import subprocess
    report = {"requested_nn_mode": args.nn_mode, "tries": []}
    order = [args.nn_mode] + [m for m in (6, 4) if m < args.nn_mode and args.nn_mode >= 5]
    for m in order:
        cmd = [sys.executable, os.path.abspath(__file__), "--preflight-child", "--preflight-device", str(device), "--nn-mode", str(m),
               "--lcp-mode", str(args.lcp_mode), "--verify-mode", str(args.verify_mode)]
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=float(os.environ.get("HOP_BENCH_PREFLIGHT_TIMEOUT_S", "300")))
            line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""
            res = json.loads(line) if r.returncode == 0 and line.startswith("{") else {"ok": False, "exit": r.returncode, "stderr_tail": r.stderr[-400:]}
        except subprocess.TimeoutExpired:
            res = {"ok": False, "error": "timeout (hang?)"}
        except Exception as e:
            res = {"ok": False, "error": f"{type(e).__name__}: {e}"}
        res["nn_mode"] = m
        res["seconds"] = time.perf_counter() - t0
        report["tries"].append(res)
        if res.get("ok"):
            report["timed_nn_mode"] = m
            return m, report
    report["timed_nn_mode"] = args.nn_mode   # nothing survived: run the requested mode and let the failure show in the parent
    return args.nn_mode, report


CFG = {"hand_match": {"finger1_min_match": 5, "finger2_min_match": 5, "finger1_dist_thres": 0.005, "finger2_dist_thres": 0.005,
                      "finger1_normal_angle": 60, "finger2_normal_angle": 60, "check_normal": True, "max_outter_pts": 300,
                      "outter_pt_dist": 0.002, "outter_pt_dist_weight": 1, "planar_dist_thres": 0.001,
                      "pso": {"n_pop": 15, "n_gen": 3, "check_freq": 10, "pso_par_c_cog": 0.1, "pso_par_c_soc": 0.9,
                              "pso_par_initial_w": 0.0}}}
TRUE_ANGLES = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}


class Workload:
    """Host-side inputs of the frame (shared) and one device context per frame in flight."""

    def __init__(self, args, rank):
        import hop_loader
        self.hop = hop_loader.load()
        from hop_amd import api
        self.api = api
        synth = self.hop.synth
        self.args = args
        self.rank = rank
        self.sc = synth.make_scene(args.scene, seed=7)
        self.model = synth.ellipsoid_model(args.model)
        self.keys = synth.ppf_key_table()
        self.hand = synth.t42_hand()
        self.hxyz, self.hnrm = synth.make_hand_scene(self.hand, TRUE_ANGLES, args.hand_scene, seed=5)
        self.swivel = self.hxyz[self.hxyz[:, 0] < -0.1]
        self.slots = []

    def setup_device(self, device, n_slots=1):
        api = self.api
        CFG["hand_match"]["pso"]["n_pop"] = self.args.particles
        for _ in range(n_slots):
            c = api.Context(device)
            c.hand_set_sum_mode(self.args.pso_sum_mode)
            c.set_scene(self.sc.xyz, self.sc.nrm, self.sc.conf, 0.8)
            c.set_model(api.HOP_MODEL_5MM, *self.model)
            c.set_model(api.HOP_MODEL_1MM, *self.model)
            c.set_ppf_keys(self.keys)
            h = api.HandT42(CFG, self.hand, ctx=c)
            h.gripper_min_dist = 0.0144
            h.setCurScene(self.hxyz, self.hnrm, self.swivel)
            opts = c.default_s4pcs_opts(sample_size=100, success_quadrilaterals=self.args.bases, max_time_seconds=0,
                                        n_trials=self.args.bases, random_seed=5489 + self.rank, verify_mode=self.args.verify_mode)
            self.slots.append(dict(ctx=c, hand=h, opts=opts))
        self.ctx = self.slots[0]["ctx"]

    def setup_device_strong(self, device, rank, world):
        """C5: H = 65 536 replay poses (seed 13) x 50 000-point scene x 5 000-point model; rank r owns the
        contiguous block [r H / N, (r + 1) H / N)."""
        api, synth = self.api, self.hop.synth
        self.sc = synth.make_scene(self.args.strong_scene, seed=13)
        H = self.args.strong_hyps
        poses = synth.replay_poses(self.sc.gt_pose, H, seed=13, max_rot_deg=30.0, max_trans=0.015)
        per = (H + world - 1) // world
        self.h0 = rank * per
        self.shard = np.ascontiguousarray(poses[self.h0:min(H, self.h0 + per)])
        c = api.Context(device)
        c.set_scene(self.sc.xyz, self.sc.nrm, self.sc.conf, 0.0)
        c.set_model(api.HOP_MODEL_5MM, *self.model)
        c.set_model(api.HOP_MODEL_1MM, *self.model)
        self.slots.append(dict(ctx=c, hand=None, opts=None))
        self.ctx = c

    def step_strong(self, slot=0, topk=0, id_offset=0, rows_dev=None):
        """One pass over this rank's block of the fixed hypothesis set: refineByICP + selectBest on all of it, top-k table."""
        c = self.slots[slot]["ctx"]
        tf = time.perf_counter()
        c.hypos_upload(self.shard)          # ICP refines the resident poses in place: every step starts from the replay set
        t2 = time.perf_counter()
        it, _ = c.icp_refine(10, 45.0, 0.01, nn_mode=self.args.nn_mode, want_stats=True)
        t3 = time.perf_counter()
        best, score, idx = c.lcp_select_best(0.001, 10.0, self.args.lcp_mode)
        t4 = time.perf_counter()
        if rows_dev is not None:
            c.topk_pack_device(topk, self.h0, rows_dev)
            rows = None
        else:
            rows = c.topk_pack(topk, id_offset=self.h0)[0] if topk > 0 else None
        return dict(h=len(self.shard), h_gen=0, n_cand=0, n_bases=0, best=best, score=score, icp_hyp_iters=int(np.sum(it)), n_pairs=0, n_quads=0,
                    rows=rows, t_frame=t2 - tf, t_pso=0.0, t_gen=0.0, t_icp=t3 - t2, t_lcp=t4 - t3, ms_select=0.0)

    def hand_search(self, h):
        for name in h._tf_self:
            h._tf_self[name] = np.eye(4, dtype=np.float32)
        # order of main_realdata_auto.cpp:114-139 (camera on the finger-2 side)
        m1 = h.matchOneComponentPSO("finger_2_1", 0, 120, False, 0.005, 60, 5)
        if m1:
            h.matchOneComponentPSO("finger_2_2", 0, 90, True, 0.005, 60, 5)
        m2 = h.matchOneComponentPSO("finger_1_1", 0, 120, False, 0.005, 60, 5)
        if m2:
            h.matchOneComponentPSO("finger_1_2", 0, 90, True, 0.005, 60, 5)
        return m1, m2

    def step(self, slot=0, topk=0, id_offset=0, rows_dev=None):
        """One frame on one context.  Returns the stage times and, if topk > 0, the packed top-k table."""
        S = self.slots[slot]
        c, h = S["ctx"], S["hand"]
        tf = time.perf_counter()
        # a new frame arrives: both clouds are handed over again, so every per-frame structure derived from them
        # (Morton order, voxel grids, NN cell lists of the scene, Verify lists, hand-scene lists) is rebuilt inside the step
        c.set_scene(self.sc.xyz, self.sc.nrm, self.sc.conf, 0.8)
        h.setCurScene(self.hxyz, self.hnrm, self.swivel)
        t0 = time.perf_counter()
        self.hand_search(h)
        t1 = time.perf_counter()
        _, _, st = c.s4pcs_generate(S["opts"], download=False)
        t2 = time.perf_counter()
        c.hypos_keep_topk(self.args.hyps)
        hh = c.hypos_count()
        it, _ = c.icp_refine(10, 45.0, 0.01, nn_mode=self.args.nn_mode, want_stats=True)
        t3 = time.perf_counter()
        best, score, idx = c.lcp_select_best(0.001, 10.0, self.args.lcp_mode)
        t4 = time.perf_counter()
        if rows_dev is not None:     # multi-GPU with the library's communicator: the table stays on the device (hop_topk_pack_device)
            c.topk_pack_device(topk, id_offset, rows_dev)
            rows = None
        else:
            rows = c.topk_pack(topk, id_offset=id_offset)[0] if topk > 0 else None
        return dict(h=hh, h_gen=st.n_hypotheses, n_cand=st.n_candidates, n_bases=st.n_bases, best=best, score=score,
                    icp_hyp_iters=int(np.sum(it)), n_pairs=st.n_pairs, n_quads=st.n_quads, rows=rows,
                    t_frame=t0 - tf, t_pso=t1 - t0, t_gen=t2 - t1, t_icp=t3 - t2, t_lcp=t4 - t3, ms_select=st.ms_select)


def _load_oracle():
    """TEST INFRASTRUCTURE: only the cpu_baseline legs import it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    return orc


def physics_row(w, args, with_cpu):
    """SURVEY.md 8(f) N1, measured on its own after the timed region (it does not enter `value`): hypotheses within
    8 deg / 4 mm of a grasp of the ellipsoid (so that most of them pass the early checks and reach the mesh checks)
    through hop_reject_by_collision; the oracle on a bounded sample of the same hypotheses beside it."""
    synth = w.hop.synth
    p, poses = synth.physics_case(args.physics_hyps, seed=21, n_model=args.model, n_scene=args.scene, mesh_subdiv=4, spacing=0.003,
                                  max_rot_deg=8.0, max_trans=0.004)
    c = w.ctx
    t0 = time.perf_counter()
    for mid, V, Fi, T in p["meshes"]:
        c.sdf_register_mesh(mid, V, Fi, T)
    ms_register = 1e3 * (time.perf_counter() - t0)
    ms_frame, ms_reject, wall = [], [], []
    keep = None
    for it in range(6):
        c.physics_set_frame(p)
        c.hypos_upload(poses)
        t0 = time.perf_counter()
        keep, diag = c.reject_by_collision()
        wall.append(1e3 * (time.perf_counter() - t0))
        a, b = c.physics_timing()
        ms_frame.append(a), ms_reject.append(b)
    ms_frame, ms_reject, wall = ms_frame[1:], ms_reject[1:], wall[1:]
    H = len(poses)
    out = {"row": "N1 rejectByCollisionOrNonTouching (PoseEstimator.cpp:524-735)",
           "hypotheses": H, "object_faces": int(len(p["object_F"])), "finger_faces": [int(len(f)) for f in p["finger_F"]],
           "finger_points": [int(len(x)) for x in p["finger_xyz"]], "model_points": int(len(p["model"])),
           "scene_points": int(len(p["cloud_without_hand"])),
           "kept": int(keep.sum()), "decided_by_check": np.bincount(diag[:, 0].astype(int), minlength=6).tolist(),
           "device_ms_reject": float(np.mean(ms_reject)), "device_ms_set_frame": float(np.mean(ms_frame)), "wall_ms_reject": float(np.mean(wall)),
           "host_ms_register_meshes": ms_register,
           "value": H / (float(np.mean(wall)) * 1e-3), "unit": "hypotheses/s",
           "bound": "VALU issue and L1/L2 latency of a per-query tree walk (meshes are cache resident: no HBM or MFMA roofline applies)"}
    if with_cpu:
        orc = _load_oracle()
        n = min(H, 64)
        t0 = time.perf_counter()
        ko, _ = orc.reject_by_collision(p, poses[:n])
        dt = time.perf_counter() - t0
        while dt < 2.0 and n < H:   # grow the sample until it takes a couple of seconds
            n = min(H, n * 4)
            t0 = time.perf_counter()
            ko, _ = orc.reject_by_collision(p, poses[:n])
            dt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": n / dt, "unit": "hypotheses/s", "cores": os.cpu_count(), "kind": "port",
                               "sample": f"first {n} of the {H} hypotheses, oracle (libigl restatement, OpenMP over hypotheses)",
                               "decisions_equal": bool(np.array_equal(ko, keep[:n]))}
    return out


def host_cpp_leg(w):
    """The C++ host above the C-ABI (icra20-hand-object-pose_amd/host/app/main_realdata_auto.cpp, the reference driver's call
    order) on the same clouds: the AS-SHIPPED chain of config_autodataset.yaml (15+1 particles x 4 fingers, <= 30 base trials
    with early stop, cluster, ICP on <= 100 hypotheses, cluster, selectBest), one frame at a time, wall clock per frame as the
    application measures it.  Separate from `value` (different hypothesis counts); it shows the cost of a frame driven
    from C++ instead of the Python mirrors."""
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib", "main_realdata_auto")
    cfg = os.path.join(ROOT, "icra20-hand-object-pose_amd", "config", "config_autodataset.yaml")
    if not os.path.exists(exe):
        return {"error": "host application not built"}

    def wc(path, xyz, nrm=None, conf=None):
        xyz = np.asarray(xyz, np.float32)
        nrm = np.zeros_like(xyz) if nrm is None else np.asarray(nrm, np.float32)
        with open(path, "wb") as f:
            np.array([len(xyz), 0 if conf is None else 1], np.int32).tofile(f)
            np.ascontiguousarray(xyz.T).tofile(f)
            np.ascontiguousarray(nrm.T).tofile(f)
            if conf is not None:
                np.asarray(conf, np.float32).tofile(f)

    with tempfile.TemporaryDirectory() as d:
        wc(os.path.join(d, "model.bin"), *w.model)
        wc(os.path.join(d, "model001.bin"), *w.model)
        wc(os.path.join(d, "object_segment.bin"), w.sc.xyz, w.sc.nrm, w.sc.conf)
        with open(os.path.join(d, "ppf_keys.bin"), "wb") as f:
            np.array([len(w.keys)], np.int32).tofile(f)
            np.ascontiguousarray(w.keys, np.int32).tofile(f)
        with open(os.path.join(d, "hand.txt"), "w") as f:
            for name in w.hand.clouds:
                if name == "base_link":
                    continue
                x, n = w.hand.clouds[name]
                wc(os.path.join(d, f"{name}.bin"), x, n)
                f.write(f"{name} {w.hand.parents[name]} {name}.bin " + " ".join(repr(float(v)) for v in w.hand.tf_in_parent[name].reshape(16)) + "\n")
        wc(os.path.join(d, "hand_scene.bin"), w.hxyz)
        wc(os.path.join(d, "hand_region.bin"), w.hxyz, w.hnrm)
        wc(os.path.join(d, "hand_swivel.bin"), w.swivel)
        open(os.path.join(d, "cam_side.txt"), "w").write("1\n")
        env = dict(os.environ, HOP_APP_REPEAT="6")
        r = subprocess.run([exe, cfg, d, d], capture_output=True, text=True, timeout=600, env=env)
        ms = [float(ln.split()[1]) for ln in r.stdout.splitlines() if ln.startswith("frame_ms")]
        if r.returncode != 0 or len(ms) < 2:
            return {"error": f"rc {r.returncode}: {(r.stdout + r.stderr)[-300:]}"}
        return {"what": "as-shipped chain (config_autodataset.yaml) driven by host/app/main_realdata_auto (C++ over the C-ABI), same C2 clouds, "
                        "one frame at a time; first pass (model-side lists built) excluded",
                "frame_ms_median": float(np.median(ms[1:])), "frame_ms_all": ms}


def _cpu_identity():
    """CPU model string and physical core count as lscpu reports them (from /proc/cpuinfo)."""
    model, cores = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1))


def cpu_baseline(w, budget_s):
    """the oracle ("port": kd-tree NN, the reference's algorithms incl. its Levenberg-Marquardt ICP minimiser) on a
    BOUNDED sample of the same workload, one warm-up then the median of 5, at 1 thread and at all physical cores; R_gen (hypotheses the
    generator emits incl. Verify per second), R_icp (hypothesis-iterations per second), R_lcp (hypotheses scored per second) separately
    and the end-to-end rate H / (t_gen + t_icp + t_lcp) with the same H through all three stages (the GPU line's accounting)."""
    import ctypes
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    model_name, phys = _cpu_identity()
    logical = os.cpu_count() or 1
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
    except OSError:
        gomp = None

    def threads(n):
        if gomp is not None:
            gomp.omp_set_num_threads(int(n))

    sc, (mx, mn) = w.sc, w.model
    keep = sc.conf >= 0.8
    S, Sn = sc.xyz[keep], sc.nrm[keep]
    H, B = w.args.hyps, w.args.bases
    t_start = time.perf_counter()

    def med5(fn):
        fn()  # warm-up
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r = fn()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), r

    # generator: the reference's base loop is sequential (one thread), so is the oracle's
    threads(1)
    n_bases_s = 6
    def gen():
        oo = orc.OracleS4PCS(sample_size=100, success_quadrilaterals=n_bases_s, n_trials=n_bases_s)
        oo.set_keys(w.keys)
        n = oo.run(sc.xyz, sc.nrm, sc.conf, mx, mn, 1)
        return n, oo.hypos()
    t_gen, (n_gen, (pose, lcp)) = med5(gen)
    order = np.argsort(-lcp, kind="stable")
    out = {"unit": "hypotheses/s", "kind": "port", "cpu_model": model_name, "physical_cores": phys, "logical_cpus": logical,
           "protocol": "one warm-up, median of 5 (BASELINE.md section 3); ICP = Utils::runICP with PCL's Levenberg-Marquardt minimiser (oracle restatement, float)",
           "R_gen_hyp_per_s_1thread": n_gen / t_gen, "runs": {}}
    for label, nt in (("1_thread", 1), ("all_physical_cores", phys)):
        threads(nt)
        n_h = int(min(len(pose), max(4, 2 * nt)))
        p = np.ascontiguousarray(pose[order[:n_h]])
        t_icp, (p2, it, cv) = med5(lambda: orc.icp_refine_batch_lm(S, Sn, mx, mn, p, 10, 45.0, 0.01))
        t_lcp, _ = med5(lambda: orc.compute_lcp_batch(sc.xyz, sc.nrm, mx, mn, p2, 0.001, 10.0, use_tree=True))
        t_frame = t_gen * (B / n_bases_s) + H * (t_icp + t_lcp) / n_h
        out["runs"][label] = {"threads": nt, "hypotheses_in_sample": n_h, "R_icp_hyp_iters_per_s": float(np.sum(it)) / t_icp, "R_lcp_hyp_per_s": n_h / t_lcp,
                              "t_gen_s": t_gen, "t_icp_s": t_icp, "t_lcp_s": t_lcp, "end_to_end_hyp_per_s": H / t_frame}
        if time.perf_counter() - t_start > 3 * budget_s:
            break
    threads(logical)
    best = out["runs"].get("all_physical_cores") or out["runs"]["1_thread"]
    out["value"] = best["end_to_end_hyp_per_s"]
    out["cores"] = best["threads"]
    out["sample"] = (f"{n_bases_s} of {B} base trials ({n_gen} hypotheses) single-thread as the reference's base loop; ICP + computeLCP on the "
                     f"{best['hypotheses_in_sample']} best of them with {best['threads']} OpenMP threads; each stage the median of 5 after a warm-up; "
                     f"end-to-end = H / (t_gen B/{n_bases_s} + H (t_icp + t_lcp) / {best['hypotheses_in_sample']}) with H = {H}")
    out["what_cannot_be_timed"] = "the reference's own PCL / Armadillo path (libraries absent); its OpenGR generator was timed in the build container: profiles/r03_ref_generator_c2.json"
    # kind "reference" for the one stage whose reference code builds: the reference's OWN generator (its OpenGR fork compiled in place in
    # the build container, g++ -O2, oracle/_ref/libref_s4pcs_o2.so travels with the snapshot) on the same C2 clouds on this box's host
    # cores, one thread, ONE ComputeTransformation call = its 30 base trials (congruentSetExplorationBase.hpp:90-100), beside the port.
    o2 = os.path.join(ROOT, "oracle", "_ref", "libref_s4pcs_o2.so")
    if os.path.exists(o2):
        saved = (orc.REF_SO, orc._ref)
        try:
            threads(1)
            orc.REF_SO, orc._ref = o2, None
            P, Pn, Pc = sc.xyz[keep], sc.nrm[keep], sc.conf[keep]

            def ref_gen():
                r = orc.RefS4PCS(success_quadrilaterals=10 ** 6, record_pairs=False, plain=True)
                r.set_keys(w.keys)
                with orc.quiet_stdout():
                    return r.run(P, Pn, Pc, mx, mn, 1)

            def port_gen():
                oo = orc.OracleS4PCS(success_quadrilaterals=10 ** 6)
                oo.set_keys(w.keys)
                return oo.run(P, Pn, Pc, mx, mn, 1)
            t_ref, n_ref = med5(ref_gen)
            t_port, n_port = med5(port_gen)
            out["reference_generator"] = {"kind": "reference", "what": "the reference's OpenGR matcher (g++ -O2 build of its own sources), untouched, one thread, 30 base trials, same clouds",
                                          "hypotheses": int(n_ref), "median_s": t_ref, "R_gen_hyp_per_s": n_ref / t_ref,
                                          "port_same_inputs": {"hypotheses": int(n_port), "median_s": t_port, "R_gen_hyp_per_s": n_port / t_port},
                                          "same_hypothesis_count": int(n_ref) == int(n_port)}
        except Exception as e:  # the line must survive a box where the prebuilt file does not load
            out["reference_generator"] = {"kind": "reference", "failed": f"{type(e).__name__}: {e}"}
        finally:
            orc.REF_SO, orc._ref = saved
            threads(logical)
    return out


def pose_vs_ground_truth(best, gt, synth, name="ellipse"):
    """(translation mm, rotation deg) of a returned pose against the frame's ground truth, modulo the object's symmetry group"""
    best, gt = np.asarray(best, np.float64).reshape(4, 4), np.asarray(gt, np.float64).reshape(4, 4)
    dr = 180.0
    for Sy in synth.symmetry_rotations(name):
        c = (np.trace(best[:3, :3].T @ (gt[:3, :3] @ Sy)) - 1.0) / 2.0
        dr = min(dr, math.degrees(math.acos(max(-1.0, min(1.0, float(c))))))
    return 1e3 * float(np.linalg.norm(best[:3, 3] - gt[:3, 3])), dr


def parity_sample(w, args, n_sample):
    """BASELINE.md 3.6 ("before any timing counts"): one more frame after the timed region, the same stages with the same seed as a timed step
    (generate -> keep the H best -> refineByICP), with the poses read back before and after the refinement; the oracle (TEST INFRASTRUCTURE,
    the checker) refines `n_sample` of them -- spread over the whole set, i.e. over every hypothesis batch of the launch -- from the same
    inputs and the refined poses are compared bit for bit (nn_mode 7 promises exactly that) and in mm / degrees."""
    c, S = w.slots[0]["ctx"], w.slots[0]
    c.set_scene(w.sc.xyz, w.sc.nrm, w.sc.conf, 0.8)
    c.s4pcs_generate(S["opts"], download=False)
    c.hypos_keep_topk(args.hyps)
    p_in, _, _ = c.hypos_download()
    p_in = p_in.copy()
    it, cv = c.icp_refine(10, 45.0, 0.01, nn_mode=args.nn_mode, want_stats=True)
    p_out, _, _ = c.hypos_download()
    H = len(p_in)
    idx = np.unique(np.linspace(0, H - 1, min(n_sample, H)).astype(int))
    orc = _load_oracle()
    keep = w.sc.conf >= 0.8
    mx, mn = w.model
    t0 = time.perf_counter()
    po, ito, cvo = orc.icp_refine_batch_lm(w.sc.xyz[keep], w.sc.nrm[keep], mx, mn, np.ascontiguousarray(p_in[idx]), 10, 45.0, 0.01, moment=(args.nn_mode == 7))
    dt = time.perf_counter() - t0
    po = np.ascontiguousarray(po, np.float32)
    pg = np.ascontiguousarray(p_out[idx], np.float32)
    eq = np.all(pg.reshape(len(idx), -1).view(np.int32) == po.reshape(len(idx), -1).view(np.int32), axis=1)
    dmm = 1e3 * np.linalg.norm(pg[:, :3, 3] - po[:, :3, 3], axis=1)
    cosang = np.clip((np.einsum("hij,hij->h", pg[:, :3, :3].astype(np.float64), po[:, :3, :3].astype(np.float64)) - 1.0) / 2.0, -1.0, 1.0)
    ddeg = np.degrees(np.arccos(cosang))
    return {"what": f"{len(idx)} of the {H} hypotheses of one more frame (same seed and stages as a timed step), indices spread over the whole set; "
                    f"refined by the oracle ({'minimiser 7: moment form' if args.nn_mode == 7 else 'float Levenberg-Marquardt'}) from the same input poses",
            "hypotheses": int(len(idx)), "bit_equal": bool(eq.all()), "bit_equal_count": int(eq.sum()),
            "iterations_equal": bool(np.array_equal(it[idx], ito)), "converged_flags_equal": bool(np.array_equal(cv[idx], cvo)),
            "max_translation_diff_mm": float(dmm.max()), "max_rotation_diff_deg": float(ddeg.max()),
            "within_1mm_1deg": int(np.sum((dmm <= 1.0) & (ddeg <= 1.0))), "oracle_seconds": dt}


def main():
    args = parse()
    if args.preflight_child:
        return preflight_child(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    preflight_report = None
    if not args.no_preflight and not os.environ.get("HOP_TEST_EMU"):
        # before anything of this process touches the device: the small frame in a child (every rank on its own device; the ranks agree on the
        # lowest surviving mode below, once the process group is up)
        args.nn_mode, preflight_report = preflight(args, local_rank)
    import threading
    import torch
    dist = None
    # HOP_BENCH_FORCE_DIST=1 runs the collective path with a single rank (self-test of the multi-GPU code on one GPU)
    use_dist = world > 1 or bool(os.environ.get("HOP_BENCH_FORCE_DIST"))
    # HOP_BENCH_BACKEND=gloo (self-test only): ranks may then share one GPU; the table travels through host memory
    backend = os.environ.get("HOP_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend != "nccl":
        local_rank = local_rank % max(ndev, 1)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    xdev = dev if backend == "nccl" else torch.device("cpu")   # where the collectives run

    if use_dist and preflight_report is not None:
        tm_ = torch.tensor([args.nn_mode], dtype=torch.int32, device=xdev)
        dist.all_reduce(tm_, op=dist.ReduceOp.MIN)   # (7 > 6 > 4: the ranks time the same arithmetic)
        args.nn_mode = int(tm_.item())
        preflight_report["timed_nn_mode"] = args.nn_mode
    strong = args.scaling == "strong"
    F = 1 if strong else max(1, args.inflight)
    w = Workload(args, rank)
    if strong:
        w.setup_device_strong(local_rank, rank, world)
        w.step = w.step_strong
    else:
        w.setup_device(local_rank, F)
    api = w.api
    K = 128  # rows of the exchanged top-k table

    # The exchange runs inside the library (hop_comm_*: ncclAllGather of the k x 72-byte table from device buffers + the
    # merge) on one RCCL communicator per rank.  Whether it is used is decided collectively -- rank 0's unique id travels
    # with a validity byte, then every rank reports whether its communicator came up -- so that a rank that cannot load
    # RCCL takes everybody to the torch.distributed exchange instead of leaving the others waiting.
    comm, exchange_kind = None, ("none" if not use_dist else "torch.distributed all_gather + hop_topk_merge")
    if use_dist and backend == "nccl" and not os.environ.get("HOP_BENCH_NO_LIB_COMM"):
        idt = torch.zeros(129, dtype=torch.uint8, device=dev)
        if rank == 0:
            try:
                idt = torch.tensor([1] + list(api.Comm.unique_id()), dtype=torch.uint8, device=dev)
            except Exception as e:
                print(f"[bench] hop_comm_unique_id failed: {e}", file=sys.stderr)
        dist.broadcast(idt, 0)
        ok = 0
        if int(idt[0].item()) == 1:
            try:
                comm = api.Comm(local_rank, bytes(idt[1:].cpu().numpy().tolist()), rank, world)
                ok = 1
            except Exception as e:
                print(f"[bench] rank {rank}: hop_comm_create failed: {e}", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            # self-check before anything is timed: the communicator RCCL built spans exactly WORLD_SIZE ranks (a line whose exchange ran on
            # fewer ranks than the launch asked for would misreport the job)
            nr = comm.info()[0]
            if nr != world:
                raise SystemExit(f"bench.py: ncclCommCount reports {nr} ranks on rank {rank}, WORLD_SIZE is {world}: refusing to time a partial job "
                                 f"(check HSA_ENABLE_IPC_MODE_LEGACY=0 and that every rank owns its own GPU)")
            exchange_kind = "hop_topk_pack_device + hop_topk_allgather_device (RCCL inside libhop.so; table on the device end to end)"
        elif comm is not None:
            comm.close()
            comm = None

    def exchange(rows, rows_dev=None):
        if not use_dist:
            return rows
        if comm is not None:
            if rows_dev is not None:
                return comm.topk_allgather_device(rows_dev, K)[0]   # device table -> ncclAllGather -> merge kernel -> 9 KB to the host
            return comm.topk_allgather(rows, K)[0]
        t = torch.from_numpy(rows).to(xdev)
        out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=xdev)
        dist.all_gather_into_tensor(out, t)
        merged, _ = api.topk_merge(out.cpu().numpy(), K)
        return merged

    def barrier():
        for S in w.slots:
            S["ctx"].synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    def run_frames(n):
        """n frames, F in flight: worker s runs frames s, s+F, ... on its own context (host selection of one frame
        overlaps device work of the others); the main thread performs the per-frame exchange in frame order."""
        infos = [None] * n
        ready = [threading.Event() for _ in range(n)]
        errors = []
        # one device table per frame of this call (9 KB each): the worker packs into it, the main thread all-gathers from it in frame order
        tables = torch.empty((n, K, api.TOPK_ROW_FLOATS), dtype=torch.float32, device=dev) if comm is not None else None
        ptr = (lambda f: tables[f].data_ptr()) if tables is not None else (lambda f: None)

        def worker(slot):
            try:
                for f in range(slot, n, F):
                    infos[f] = w.step(slot, K, rank * (1 << 24), rows_dev=ptr(f))
                    ready[f].set()
            except BaseException as e:  # surface the failure in the main thread
                errors.append(e)
                for ev in ready:
                    ev.set()
                if use_dist:
                    # the other ranks are (or will be) inside the per-frame collective: take the whole job down instead of
                    # leaving them blocked until the RCCL timeout
                    import traceback
                    traceback.print_exc()
                    sys.stderr.flush()
                    os._exit(3)

        threads = [threading.Thread(target=worker, args=(s,)) for s in range(min(F, n))]
        for t in threads:
            t.start()
        table = None
        for f in range(n):
            ready[f].wait()
            if errors:
                break
            table = exchange(infos[f]["rows"], ptr(f))
        for t in threads:
            t.join()
        if errors:
            raise errors[0]
        return infos, table

    if args.warmup > 0:
        run_frames(F * ((args.warmup + F - 1) // F))   # every context sees at least one warm-up frame
    for S in w.slots:
        S["ctx"].timing_enable(True)
        S["ctx"].timing_reset()
    barrier()
    t0 = time.perf_counter()
    infos, table = run_frames(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    tm = {}
    for S in w.slots:
        for k, v in S["ctx"].timing_get().items():
            tm[k] = tm.get(k, 0) + v
        S["ctx"].timing_enable(False)

    # With several frames in flight the event spans of one frame's kernels include time spent sharing the device with
    # the other frame's kernels.  One more frame is therefore run ALONE, after the timed region (it does not enter
    # `value`), to measure every kernel undisturbed; both measurements are reported.
    tm_serial, info_serial = None, None
    if F > 1 and not args.no_serial_frame:
        c0 = w.slots[0]["ctx"]
        c0.timing_enable(True)
        c0.timing_reset()
        serial_table = torch.empty((K, api.TOPK_ROW_FLOATS), dtype=torch.float32, device=dev) if comm is not None else None
        info_serial = w.step(0, K, rank * (1 << 24), rows_dev=serial_table.data_ptr() if serial_table is not None else None)
        c0.synchronize()
        tm_serial = c0.timing_get()
        c0.timing_enable(False)
        if use_dist:
            exchange(info_serial["rows"], serial_table.data_ptr() if serial_table is not None else None)   # keep the collective sequence identical on every rank

    # The same workload in the other configurations, measured after the timed region (they do not enter `value`): ICP with one
    # Gauss-Newton step per iteration instead of the reference's Levenberg-Marquardt minimiser, and the configuration in which
    # every stage is bit-equal to the oracle's operation order (chained ICP increments, ordered computeLCP / objFuncPSO sums).
    alt = {}
    if not strong and not args.no_alt_modes:
        base_cfg = (args.nn_mode, args.lcp_mode, args.pso_sum_mode)
        for label, cfg in (("icp_float_moment_sums_nn_mode6", (6, args.lcp_mode, args.pso_sum_mode)),
                           ("icp_one_gauss_newton_step_nn_mode4", (4, args.lcp_mode, args.pso_sum_mode)),
                           ("oracle_bits_every_stage_nn_mode7_lcp_mode2_pso_sum_mode0", (7, 2, 0))):
            if cfg == base_cfg:
                continue
            args.nn_mode, args.lcp_mode, args.pso_sum_mode = cfg
            for S in w.slots:
                S["ctx"].hand_set_sum_mode(args.pso_sum_mode)
            run_frames(F)
            barrier()
            ta = time.perf_counter()
            ai, _ = run_frames(2 * F)
            barrier()
            dt = time.perf_counter() - ta
            hl = float(sum(i["h"] for i in ai))
            if use_dist:
                tt = torch.tensor([hl], dtype=torch.float64, device=xdev)
                dist.all_reduce(tt, op=dist.ReduceOp.SUM)
                td = torch.tensor([dt], dtype=torch.float64, device=xdev)
                dist.all_reduce(td, op=dist.ReduceOp.MAX)
                hl, dt = float(tt.item()), float(td.item())
            alt[label] = {"value": hl / dt, "unit": "hypotheses/s", "ms_per_step": 1e3 * dt / (2 * F), "steps": 2 * F,
                          "nn_mode": cfg[0], "lcp_mode": cfg[1], "pso_sum_mode": cfg[2],
                          "best_pose_max_abs_diff_vs_default": float(np.abs(ai[-1]["best"] - infos[-1]["best"]).max())}
        args.nn_mode, args.lcp_mode, args.pso_sum_mode = base_cfg
        for S in w.slots:
            S["ctx"].hand_set_sum_mode(args.pso_sum_mode)

    h_local = sum(i["h"] for i in infos)
    if use_dist:
        t = torch.tensor([elapsed, float(h_local)], dtype=torch.float64, device=xdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        h_total = float(t[1].item())
    else:
        h_total = float(h_local)

    if rank == 0:
        N, M = w.ctx.L.hop_scene_size(w.ctx.h), args.model
        steps = max(args.steps, 1)
        H = infos[-1]["h"]
        nq = 100
        bytes_per_hyp = 24 * (N + M) + 72                      # SoA xyz+normal of both clouds once, pose in, score out
        flops_per_hyp = 8 * N * M + 18 * M + 30 * N             # one brute-force NN pass (the reference's work per hypothesis)
        def kernel_table(tmx, frames):
            """per kernel: (device ms, launches, algorithmic bytes, brute-force-equivalent flops) over `frames`"""
            nf = len(frames)
            hyp_iters = sum(i["icp_hyp_iters"] for i in frames)     # hypothesis-iterations ICP actually ran
            n_cand = sum(i["n_cand"] for i in frames)
            kern = {
                "k_icp_corr_cells": (tmx["ms_icp_nn"], tmx["n_icp_nn_launches"], hyp_iters * bytes_per_hyp, hyp_iters * flops_per_hyp),
                "k_icp_accum": (tmx["ms_icp_accum"], tmx["n_icp_nn_launches"], hyp_iters * bytes_per_hyp, 0.0),
                "k_lcp_cells": (tmx["ms_lcp_fwd"] + tmx["ms_lcp_rev"], tmx["n_lcp_launches"], H * nf * bytes_per_hyp, 2.0 * H * nf * flops_per_hyp),
                "k_lcp_sum": (tmx["ms_lcp_sum"], tmx["n_lcp_launches"], H * nf * (8 * N + 4), 0.0),
                # Verify on EXIST-mode cell lists never streams P: per candidate it reads the transform (48 B), nq samples
                # (12 B each, shared through L2) and per sample one 8-byte range record + at most one 16-byte entry
                "k_verify_cells": (tmx["ms_verify"], tmx["n_verify_launches"], n_cand * (48 + nq * (12 + 8 + 16) + 4), 0.0),
                "k_quads": (tmx["ms_quads"], tmx["n_quads_launches"],
                            sum(i["n_pairs"] for i in frames) * 8 + sum(i["n_quads"] for i in frames) * 36 + n_cand * 88, 0.0),
                "k_ppf_matrix": (tmx["ms_ppf_matrix"], nf, nf * (24 * N + N * N / 8), 0.0),
            }
            if args.nn_mode < 2:
                kern["k_icp_nn"] = kern.pop("k_icp_corr_cells")
                kern.pop("k_icp_accum")
            elif args.nn_mode >= 3:
                # nn_mode 6 / 7: lookups + the 13 x 13 moment sums in one kernel, then the whole Levenberg-Marquardt run per hypothesis
                # which of the two moment kernels ran: the library says (HOP_ICP_MFMA=0, or a device that failed the read-out check, take the vector units)
                eng = w.ctx.L.hop_debug_icp_engine(w.ctx.h) if hasattr(w.ctx.L, "hop_debug_icp_engine") else -1
                momi = "k_icp_fusedq_momi" if (eng == 0 or (eng < 0 and os.environ.get("HOP_ICP_MFMA", "1") == "0")) else "k_icp_fusedq_momm"
                kern[{6: "k_icp_fusedq_mom", 7: momi}.get(args.nn_mode, "k_icp_fusedq")] = kern.pop("k_icp_corr_cells")
                kern.pop("k_icp_accum")
                if args.nn_mode >= 5:
                    kern[{6: "k_icp_lm6_solve", 7: "k_icp_lm7_solve"}.get(args.nn_mode, "k_icp_lm_pass+solve")] = (tmx["ms_icp_solve"], tmx["n_icp_nn_launches"], 0.0, 0.0)
            return kern

        def roofline_of(kern, dom, frames):
            ms_dom, n_dom, alg_bytes_total, alg_flops_total = kern[dom]
            avg_ms = ms_dom / max(n_dom, 1)
            alg_bytes = alg_bytes_total / max(n_dom, 1)
            ach_gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
            ach_tf = alg_flops_total / max(n_dom, 1) / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
            nf = max(len(frames), 1)
            return {"kernel": dom, "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach_gbs / HBM_PEAK_GBS,
                    "avg_launch_ms": avg_ms, "launches": n_dom, "algorithmic_bytes_per_launch": alg_bytes,
                    "brute_force_equivalent_fp32_TFLOPs": ach_tf,
                    "per_kernel": {k: {"ms_per_frame": v[0] / nf, "launches_per_frame": v[1] / nf,
                                       "algorithmic_GBps": (v[2] / (v[0] * 1e-3) / 1e9 if v[0] > 0 else 0.0)} for k, v in kern.items()}}

        kern = kernel_table(tm, infos)
        dom = max(kern, key=lambda k: kern[k][0])
        roof = roofline_of(kern, dom, infos)
        # HBM traffic and the issue-side utilisation of the dominant kernel come from SEPARATE rocprofv3 --pmc passes (tools/gpu_profile.sh ->
        # profiles/pmc_latest.json, profiles/pmc_sq_latest.json), never from this run.  A kernel those files do not hold is said to be
        # unmeasured -- with its predecessor's figures beside it, labelled as such -- instead of a bare null.
        PREDECESSOR = {"k_icp_fusedq_momm": "k_icp_fusedq_mom", "k_icp_fusedq_momi": "k_icp_fusedq_mom"}

        def pmc_lookup(fname):
            try:
                return json.load(open(os.path.join(ROOT, "profiles", fname)))
            except Exception:
                return {}
        pmc_hbm, pmc_sq = pmc_lookup("pmc_latest.json"), pmc_lookup("pmc_sq_latest.json")
        traffic = pmc_hbm.get(dom, {}).get("hbm_bytes_per_launch")
        pmc_extra = {"traffic_of": dom if traffic is not None else None}
        if traffic is None:
            pre = PREDECESSOR.get(dom)
            pmc_extra["traffic_status"] = f"unmeasured for {dom} (no --pmc pass has run since it was written)"
            if pre and pre in pmc_hbm:
                pmc_extra["traffic_predecessor"] = {"kernel": pre, "hbm_bytes_per_launch": pmc_hbm[pre].get("hbm_bytes_per_launch"),
                                                    "note": "the kernel this one replaced, same lookups and lists; NOT this kernel's traffic"}
        sq = pmc_sq.get(dom)
        sq_of = dom
        if sq is None and PREDECESSOR.get(dom) in pmc_sq:
            sq_of = PREDECESSOR[dom] + " (predecessor; " + dom + " unmeasured)"
            sq = pmc_sq[PREDECESSOR[dom]]
        # the roof that actually binds the cell-list kernels is vector issue and L1 tag rate (a VALU-side fraction is mandatory)
        pmc_extra["valu_busy"] = sq.get("valu_busy") if sq else None
        pmc_extra["l1_accesses_per_clk_cu"] = sq.get("l1_accesses_per_clk_cu") if sq else None
        pmc_extra["issue_counters_of"] = sq_of if sq else f"unmeasured for {dom}"
        pmc_extra["issue_counters_source"] = pmc_sq.get("_source")
        roof["frac_timed_region"] = roof["frac"]
        roof = {"bound": "hbm", **roof, "traffic": traffic, **pmc_extra,
                "traffic_source": "profiles/pmc_latest.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command taken separately "
                                  "(MI355X_MICROARCH.md corrections applied by tools/pmc_summary.py), per launch; not measured in this run",
                "definition": "achieved = algorithmic bytes per launch (SURVEY.md 8(d): 24*(N+M)+72 B per hypothesis and NN pass, x the "
                              "hypotheses -- for ICP the hypothesis-iterations -- one launch processes) / mean launch duration from HIP events "
                              "on the context stream over the timed region",
                "brute_force_note": "flops the reference's brute-force NN would spend on the same units; the cell-list kernels do ~1e-3 of "
                                    "them, so brute_force_equivalent_fp32_TFLOPs (peak %.1f) is a speed-up measure, not a utilisation" % VALU_PEAK_TFLOPS}
        if tm_serial is not None:
            ks = kernel_table(tm_serial, [info_serial])
            roof["serial_frame"] = {"note": "same kernels measured on one extra frame run alone after the timed region "
                                            "(frames_in_flight > 1 stretches the spans above by cross-frame sharing of the device)",
                                    **roofline_of(ks, dom, [info_serial])}
            roof["frac_serial_frame"] = roof["serial_frame"]["frac"]
        hyp_iters = sum(i["icp_hyp_iters"] for i in infos)
        stage = lambda key: 1e3 * float(np.mean([i[key] for i in infos]))
        out = {
            "metric": "pose hypotheses/sec (gen+ICP+LCP) per frame",
            "value": h_total / elapsed,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("f32; ICP moment operands on a 13-bit fixed-point grid (|U| <= 2^12: i8 x i8 -> i32 on the matrix cores, i64 across workgroups), f64 solve"
                      if args.nn_mode == 7 else "f32"),
            "data": "synthetic",
            "config": {"workload": ("C5: fixed replay set of %d hypotheses (seed 13) x %d-pt scene / 5k-pt model, refineByICP + selectBest, split over the ranks"
                                    % (args.strong_hyps, args.strong_scene)) if strong else
                                   "C2: ellipse, 2048 Super4PCS base trials + 200-particle hand search, 20k-pt scene / 5k-pt model",
                       "exchange": exchange_kind,
                       "scene_points": N, "model_points": M, "base_trials": args.bases, "sample_size": nq,
                       "hypotheses_scored_per_rank": H, "pso_particles": args.particles, "hand_scene_points": args.hand_scene,
                       "verify_mode": args.verify_mode, "nn_mode": args.nn_mode, "lcp_mode": args.lcp_mode, "pso_sum_mode": args.pso_sum_mode, "frames_in_flight": F,
                       "parallelism": f"hypothesis-parallel x{world}, all-gather top-{K}"},
            "roofline": roof,
            "stage_ms_per_frame": {"frame_handover": stage("t_frame"), "pso": stage("t_pso"), "generate": stage("t_gen"),
                                   "generate_host_select": float(np.mean([i["ms_select"] for i in infos])),
                                   "icp": stage("t_icp"), "lcp": stage("t_lcp")},
            "frame_latency_ms": {"note": "ms_per_step is a throughput figure (frames overlap); these are per-frame latencies, hand-over to best pose",
                                 "with_frames_in_flight": float(np.mean([1e3 * (i["t_frame"] + i["t_pso"] + i["t_gen"] + i["t_icp"] + i["t_lcp"]) for i in infos])),
                                 "one_frame_alone": (1e3 * (info_serial["t_frame"] + info_serial["t_pso"] + info_serial["t_gen"] + info_serial["t_icp"]
                                                            + info_serial["t_lcp"]) if info_serial else None)},
            "device_ms_total": {k: v for k, v in tm.items() if k.startswith("ms_")},
            "icp_hypothesis_iterations_per_step": hyp_iters / steps,
            "hypotheses_generated_per_step": infos[-1]["h_gen"], "candidates_verified_per_step": infos[-1]["n_cand"],
            "best_lcp_score": infos[-1]["score"],
            "alt_modes": alt,
        }
        if preflight_report is not None:
            out["preflight"] = preflight_report
        try:
            # which moment kernel nn_mode 7 ran, and what the library's first-use checks of the gfx950-specific instructions said on this device
            # (False = the library substituted a kernel and said so on stderr: a finding about the part, not a pass)
            chk = w.ctx.selfcheck(force=False)
            out["icp_engine"] = chk["icp_engine"] if args.nn_mode == 7 else None
            out["device_selfcheck"] = {"packed_ranking_q_rank": chk["qrank"], "matrix_core_read_out": chk["mfma"]}
        except Exception as e:
            out["device_selfcheck"] = {"error": f"{type(e).__name__}: {e}"}
        if not strong:
            # the returned pose of every timed frame against the frame's ground truth, modulo the ellipsoid's symmetry (north_star: 1 mm / 1 deg
            # is asked of the pose against the CPU reference -- parity_sample below; this is the absolute error the chain ends with)
            errs = [pose_vs_ground_truth(i["best"], w.sc.gt_pose, w.hop.synth) for i in infos]
            out["best_pose_vs_ground_truth_mm"] = float(max(e[0] for e in errs))
            out["best_pose_vs_ground_truth_deg"] = float(max(e[1] for e in errs))
            out["best_pose_same_in_every_timed_frame"] = bool(all(np.array_equal(i["best"], infos[0]["best"]) for i in infos))
            if args.parity_sample > 0 and args.nn_mode >= 5 and world == 1:
                try:
                    out["parity_sample"] = parity_sample(w, args, args.parity_sample)
                except Exception as e:  # reported, never allowed to lose the line
                    out["parity_sample"] = {"error": f"{type(e).__name__}: {e}"}
        if comm is not None:
            nr, us, cnt = comm.info()
            out["config"]["rccl_ranks"] = nr
            out["exchange"] = {"mean_us": us, "count": cnt, "what": "ncclAllGather of the 128 x 72-byte table from device memory + merge kernel + 9 KB to the host, "
                                                                        "wall time per call on the main thread (the pack runs on the frame's own thread)"}
        out["config"]["icp_minimiser"] = ("Levenberg-Marquardt on (t, quaternion) to Eigen's stopping rule per ICP iteration = the reference's "
                                          "(PCL TransformationEstimationPointToPlane, Utils.cpp:200-216)" if args.nn_mode >= 5 else
                                          "one Gauss-Newton step per ICP iteration (NOT the reference's minimiser; see alt_modes / --nn-mode 6)")
        out["config"]["parity_note"] = ("ICP nn_mode 7: the reference's minimiser evaluated from integer-exact moment sums with an IEEE-only solve -- the refined "
                                        "poses are the oracle's (minimiser 7) bit for bit, whatever the launch shape; nn_mode 6 is the float-sum variant of r03. "
                                        "computeLCP lcp_mode 3 and objFuncPSO sum mode 1 re-associate float sums (<= 1e-4 / 1e-5 relative). "
                                        "alt_modes.oracle_bits_every_stage_* is the configuration in which every stage returns the oracle's bits.")
        if not args.no_cpu_baseline and world == 1 and not strong:
            try:
                out["cpu_baseline"] = cpu_baseline(w, args.cpu_budget_s)
            except Exception as e:  # the baseline is reported, never required for the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "hypotheses/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        if strong:
            args.no_next_rows = True
        if not args.no_next_rows and world == 1:
            try:
                out["host_cpp_frame"] = host_cpp_leg(w)
            except Exception as e:
                out["host_cpp_frame"] = {"error": str(e)}
        if not args.no_next_rows and world == 1:
            try:
                out["next_rows"] = {"n1_physics": physics_row(w, args, not args.no_cpu_baseline)}
            except Exception as e:
                out["next_rows"] = {"n1_physics": {"value": None, "error": str(e)}}
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL writes its version banner through C stdio, which is flushed at exit when stdout is a pipe: flush it now
        # so that the JSON line is the last line of the output
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
