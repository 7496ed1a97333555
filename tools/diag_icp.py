"""Diagnostic (GPU): ICP iteration histogram and per-call timings at C2 sizes."""
import sys, time, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import hop_loader
hop = hop_loader.load()
from hop_amd import api
synth = hop.synth
ctx = api.Context(0)
sc = synth.make_scene(20000, seed=7)
if os.environ.get('SORT_SCENE'):
    q = ((sc.xyz - sc.xyz.min(0)) / 0.002).astype(np.int64)
    def part(v):
        v = v & 0x3ff; v = (v | (v << 16)) & 0x30000ff; v = (v | (v << 8)) & 0x300f00f; v = (v | (v << 4)) & 0x30c30c3; v = (v | (v << 2)) & 0x9249249; return v
    code = part(q[:,0]) | (part(q[:,1]) << 1) | (part(q[:,2]) << 2)
    order = np.argsort(code, kind='stable')
    sc.xyz, sc.nrm, sc.conf = np.ascontiguousarray(sc.xyz[order]), np.ascontiguousarray(sc.nrm[order]), np.ascontiguousarray(sc.conf[order])
mx, mn = synth.ellipsoid_model(5000)
keys = synth.ppf_key_table()
ctx.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
ctx.set_model(api.HOP_MODEL_5MM, mx, mn)
ctx.set_model(api.HOP_MODEL_1MM, mx, mn)
ctx.set_ppf_keys(keys)
o = ctx.default_s4pcs_opts(sample_size=100, success_quadrilaterals=2048, max_time_seconds=0, n_trials=2048, verify_mode=1)
t = time.perf_counter(); _, _, st = ctx.s4pcs_generate(o, download=False); print("generate", time.perf_counter() - t, st.n_hypotheses, st.n_candidates, st.ms_select)
ctx.hypos_keep_topk(10240)
p0, s0, _ = ctx.hypos_download()
print("verify-lcp of kept set: min %.2f max %.2f" % (s0.min(), s0.max()))
for mode in (2, 2, 3):
    ctx.hypos_upload(p0, s0)
    ctx.timing_enable(True); ctx.timing_reset()
    t = time.perf_counter(); it, cv = ctx.icp_refine(10, 45.0, 0.01, nn_mode=mode, want_stats=True); dt = time.perf_counter() - t
    tm = ctx.timing_get()
    print("mode", mode, "icp wall %.1f ms  device nn %.1f ms" % (1e3 * dt, tm["ms_icp_nn"]), "iters hist", np.bincount(it, minlength=11), "mean", it.mean(), "converged", cv.mean())
t = time.perf_counter(); b = ctx.lcp_select_best(0.001, 10.0, 1); print("lcp wall", time.perf_counter() - t, "best", b[1])
