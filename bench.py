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

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s
VALU_PEAK_TFLOPS = 157.3   # FP32 vector peak, same guide


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", type=int, default=20000)
    ap.add_argument("--model", type=int, default=5000)
    ap.add_argument("--bases", type=int, default=2048)
    ap.add_argument("--hyps", type=int, default=10240)
    ap.add_argument("--particles", type=int, default=200)
    ap.add_argument("--hand-scene", type=int, default=20000)
    ap.add_argument("--verify-mode", type=int, default=2, help="0 brute-force LDS scan, 1 voxel grid, 2 EXIST-mode cell lists (identical counts)")
    ap.add_argument("--nn-mode", type=int, default=2, help="ICP / computeLCP nearest neighbour: 0 brute force, 1 voxel grid, 2 NN cell lists for ICP (identical results)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    return ap.parse_args()


CFG = {"hand_match": {"finger1_min_match": 5, "finger2_min_match": 5, "finger1_dist_thres": 0.005, "finger2_dist_thres": 0.005,
                      "finger1_normal_angle": 60, "finger2_normal_angle": 60, "check_normal": True, "max_outter_pts": 300,
                      "outter_pt_dist": 0.002, "outter_pt_dist_weight": 1, "planar_dist_thres": 0.001,
                      "pso": {"n_pop": 15, "n_gen": 3, "check_freq": 10, "pso_par_c_cog": 0.1, "pso_par_c_soc": 0.9,
                              "pso_par_initial_w": 0.0}}}
TRUE_ANGLES = {"finger_1_1": math.radians(10), "finger_1_2": math.radians(6), "finger_2_1": math.radians(12), "finger_2_2": math.radians(5)}


class Workload:
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

    def setup_device(self, device):
        api = self.api
        self.ctx = api.Context(device)
        c = self.ctx
        c.set_scene(self.sc.xyz, self.sc.nrm, self.sc.conf, 0.8)
        c.set_model(api.HOP_MODEL_5MM, *self.model)
        c.set_model(api.HOP_MODEL_1MM, *self.model)
        c.set_ppf_keys(self.keys)
        CFG["hand_match"]["pso"]["n_pop"] = self.args.particles
        self.handt42 = api.HandT42(CFG, self.hand, ctx=c)
        self.handt42.gripper_min_dist = 0.0144
        self.handt42.setCurScene(self.hxyz, self.hnrm, self.swivel)
        self.opts = c.default_s4pcs_opts(sample_size=100, success_quadrilaterals=self.args.bases, max_time_seconds=0,
                                         n_trials=self.args.bases, random_seed=5489 + self.rank, verify_mode=self.args.verify_mode)

    def hand_search(self):
        h = self.handt42
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

    def step(self):
        c = self.ctx
        tf = time.perf_counter()
        # a new frame arrives: both clouds are handed over again, so every per-frame structure derived from them
        # (Morton order, voxel grids, NN cell lists of the scene, Verify lists, hand-scene grid) is rebuilt inside the step
        c.set_scene(self.sc.xyz, self.sc.nrm, self.sc.conf, 0.8)
        self.handt42.setCurScene(self.hxyz, self.hnrm, self.swivel)
        t0 = time.perf_counter()
        self.hand_search()
        t1 = time.perf_counter()
        _, _, st = c.s4pcs_generate(self.opts, download=False)
        t2 = time.perf_counter()
        c.hypos_keep_topk(self.args.hyps)
        h = c.hypos_count()
        c.icp_refine(10, 45.0, 0.01, nn_mode=self.args.nn_mode)
        c.synchronize()
        t3 = time.perf_counter()
        best, score, idx = c.lcp_select_best(0.001, 10.0, self.args.nn_mode)
        t4 = time.perf_counter()
        return dict(h=h, h_gen=st.n_hypotheses, n_cand=st.n_candidates, n_bases=st.n_bases, best=best, score=score,
                    t_frame=t0 - tf, t_pso=t1 - t0, t_gen=t2 - t1, t_icp=t3 - t2, t_lcp=t4 - t3, ms_select=st.ms_select)


def cpu_baseline(w, budget_s):
    """The oracle ("port", kd-tree NN, OpenMP on all host cores) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    cores = os.cpu_count() or 1
    sc, (mx, mn) = w.sc, w.model
    n_bases_s = 24
    t0 = time.perf_counter()
    oo = orc.OracleS4PCS(sample_size=100, success_quadrilaterals=n_bases_s, n_trials=n_bases_s)
    oo.set_keys(w.keys)
    n_gen = oo.run(sc.xyz, sc.nrm, sc.conf, mx, mn, 1)
    t_gen = time.perf_counter() - t0
    pose, lcp = oo.hypos()
    n_h = min(len(pose), 4 * cores)
    order = np.argsort(-lcp, kind="stable")[:n_h]
    p = pose[order]
    t0 = time.perf_counter()
    p2, it, cv = orc.icp_refine_batch(sc.xyz, sc.nrm, mx, mn, p, 10, 45.0, 0.01, use_tree=True)
    t_icp = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.compute_lcp_batch(sc.xyz, sc.nrm, mx, mn, p2, 0.001, 10.0, use_tree=True)
    t_lcp = time.perf_counter() - t0
    # same accounting as the GPU line: H hypotheses through gen (all bases) + ICP + LCP
    H, B = w.args.hyps, w.args.bases
    t_frame = t_gen * (B / n_bases_s) + H * (t_icp + t_lcp) / max(n_h, 1)
    return {"value": H / t_frame, "unit": "hypotheses/s", "cores": cores, "kind": "port",
            "sample": f"oracle (kd-tree NN; generator single-thread as the reference's base loop, ICP/LCP OpenMP x{cores}): "
                      f"{n_bases_s} of {B} base trials ({n_gen} hyps, {t_gen:.2f}s), {n_h} of {H} hypotheses ICP {t_icp:.2f}s "
                      f"+ computeLCP {t_lcp:.2f}s; extrapolated to the full frame"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    w = Workload(args, rank)
    w.setup_device(local_rank)
    api = w.api
    K = 128  # rows of the exchanged top-k table

    def exchange():
        rows, n = w.ctx.topk_pack(K, id_offset=rank * (1 << 24))
        if world == 1:
            return rows
        t = torch.from_numpy(rows).to(dev)
        out = torch.empty((world * t.shape[0], t.shape[1]), dtype=t.dtype, device=dev)
        dist.all_gather_into_tensor(out, t)
        merged, _ = api.topk_merge(out.cpu().numpy(), K)
        return merged

    def barrier():
        w.ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        w.step()
        exchange()
    w.ctx.timing_enable(True)
    w.ctx.timing_reset()
    barrier()
    t0 = time.perf_counter()
    infos = []
    for _ in range(args.steps):
        infos.append(w.step())
        table = exchange()
    barrier()
    elapsed = time.perf_counter() - t0
    tm = w.ctx.timing_get()
    w.ctx.timing_enable(False)

    h_local = sum(i["h"] for i in infos)
    if world > 1:
        t = torch.tensor([elapsed, float(h_local)], dtype=torch.float64, device=dev)
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
        # dominant kernel: the one with the largest share of device time
        kern = {
            "k_lcp_forward": (tm["ms_lcp_fwd"], tm["n_lcp_launches"]),
            "k_lcp_reverse": (tm["ms_lcp_rev"], tm["n_lcp_launches"]),
            "k_icp_nn": (tm["ms_icp_nn"], tm["n_icp_nn_launches"]),
            "k_verify": (tm["ms_verify"], tm["n_verify_launches"]),
        }
        dom = max(kern, key=lambda k: kern[k][0])
        ms_dom, n_dom = kern[dom]
        H = infos[-1]["h"]
        bytes_per_hyp = 24 * (N + M) + 72                     # SoA xyz+normal of both clouds once, pose in, score out
        flops_per_hyp_pass = 8 * N * M + 18 * M + 30 * N      # one brute-force NN pass
        if dom.startswith("k_lcp"):
            hyps_per_launch = H * steps / max(n_dom, 1)
            passes = 1.0
        elif dom == "k_icp_nn":
            hyps_per_launch = H * steps / max(n_dom, 1)        # upper bound: converged hypotheses exit early
            passes = 1.0
        else:
            hyps_per_launch = sum(i["n_cand"] for i in infos) / max(n_dom, 1)
            passes = 1.0
        avg_ms = ms_dom / max(n_dom, 1)
        if dom == "k_verify":
            nq = 100
            alg_bytes = hyps_per_launch * (12 * (N + nq) + 64 + 4)
            alg_flops = hyps_per_launch * (8.0 * N * nq)
        else:
            alg_bytes = hyps_per_launch * bytes_per_hyp
            alg_flops = hyps_per_launch * flops_per_hyp_pass * passes
        ach_gbs = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        ach_tf = alg_flops / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "pose hypotheses/sec (gen+ICP+LCP) per frame",
            "value": h_total / elapsed,
            "unit": "hypotheses/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: ellipse, 2048 Super4PCS base trials + 200-particle hand search, 20k-pt scene / 5k-pt model",
                       "scene_points": N, "model_points": M, "base_trials": args.bases, "sample_size": 100,
                       "hypotheses_scored_per_rank": H, "pso_particles": args.particles, "hand_scene_points": args.hand_scene,
                       "verify_mode": args.verify_mode, "nn_mode": args.nn_mode, "parallelism": f"hypothesis-parallel x{world}, all-gather top-{K}"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": ach_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach_gbs / HBM_PEAK_GBS, "traffic": traffic, "avg_launch_ms": avg_ms, "launches": n_dom,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "valu_fp32": {"achieved": ach_tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": ach_tf / VALU_PEAK_TFLOPS,
                                       "note": "brute-force NN is FP32-VALU bound (~1300 flop/B); both fractions reported, SURVEY.md 8(d)"}},
            "stage_ms_per_step": {"frame_handover": 1e3 * np.mean([i["t_frame"] for i in infos]), "pso": 1e3 * np.mean([i["t_pso"] for i in infos]), "generate": 1e3 * np.mean([i["t_gen"] for i in infos]),
                                  "generate_host_select": float(np.mean([i["ms_select"] for i in infos])),
                                  "icp": 1e3 * np.mean([i["t_icp"] for i in infos]), "lcp": 1e3 * np.mean([i["t_lcp"] for i in infos])},
            "device_ms_total": {k: v for k, v in tm.items() if k.startswith("ms_")},
            "hypotheses_generated_per_step": infos[-1]["h_gen"], "candidates_verified_per_step": infos[-1]["n_cand"],
            "best_lcp_score": infos[-1]["score"],
        }
        if not args.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(w, args.cpu_budget_s)
            except Exception as e:  # the baseline is reported, never required for the GPU line
                out["cpu_baseline"] = {"value": None, "unit": "hypotheses/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e}"}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
