#!/usr/bin/env python
"""Per-query statistics of the packed ICP lookups (cells_nnq) on the C2 hypothesis set: builds a second copy of the
library with -DHOP_ICP_COUNT under tools/_tmp/cnt and runs the ICP stage once.

    python tools/icp_counters.py            (on the GPU box; the counting library is built here when missing)
    python tools/icp_counters.py --build    (build only: run in the container so that the .so travels with gpurun)
"""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tools", "_tmp", "cnt")
SRC = os.path.join(ROOT, "icra20-hand-object-pose_amd", "csrc")


def build():
    os.makedirs(OUT, exist_ok=True)
    import re
    flags = ("--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -DHOP_ICP_COUNT " + re.search(r"^KERNELS_FLAGS := (.*)$", open(os.path.join(SRC, "Makefile")).read(), re.M).group(1)).split()
    subprocess.check_call(["make", "-C", SRC], stdout=subprocess.DEVNULL)
    o = os.path.join(OUT, "hop_kernels_count.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(SRC, "hop_kernels.hip"), "-o", o])
    objs = [o] + [os.path.join(SRC, "..", "lib", "obj", n + ".o") for n in ("hop_ctx", "hop_physics", "hop_normals", "hop_render", "hop_comm")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-ldl", "-o", os.path.join(OUT, "libhop_count.so")])


def main():
    if "--build" in sys.argv:
        build()
        return
    lib_path = os.path.join(OUT, "libhop_count.so")
    if not os.path.exists(lib_path):
        build()
    import hop_loader
    hop = hop_loader.load()
    from hop_amd import api
    api.LIB_PATH = lib_path
    synth = hop.synth
    sc = synth.make_scene(20000, seed=7)
    mx, mn = synth.ellipsoid_model(5000)
    c = api.Context(0)
    c.set_scene(sc.xyz, sc.nrm, sc.conf, 0.8)
    c.set_model(api.HOP_MODEL_5MM, mx, mn)
    c.set_ppf_keys(synth.ppf_key_table())
    o = c.default_s4pcs_opts(sample_size=100, success_quadrilaterals=2048, max_time_seconds=0, n_trials=2048, random_seed=5489)
    c.s4pcs_generate(o, download=False)
    c.hypos_keep_topk(10240)
    cnt = (C.c_ulonglong * 8)()
    c.L.hop_debug_icp_counters.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int]
    c.L.hop_debug_icp_counters(c.h, cnt, 1)
    it, cv = c.icp_refine(10, 45.0, 0.01, nn_mode=4, want_stats=True)
    c.L.hop_debug_icp_counters(c.h, cnt, 1)
    v = [int(x) for x in cnt]
    q = max(v[0], 1)
    print(json.dumps({"hyp_iters": int(it.sum()), "queries_in_grid": v[0], "queries_with_candidate": v[1], "chunks_per_query": v[2] / q,
                      "wave_loop_trips_x64_per_query": 64.0 * v[3] / q, "fallback_lanes_per_query": v[4] / q,
                      "waves_with_fallback_x64_per_query": 64.0 * v[5] / q, "accepted_per_query": v[6] / q,
                      "waves_with_accepted_x64_per_query": 64.0 * v[7] / q}))


if __name__ == "__main__":
    main()
