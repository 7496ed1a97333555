#!/usr/bin/env python
"""Drives the physics rejection row alone, for rocprofv3 and for the traversal counters.

    rocprofv3 --kernel-trace --stats -d gpurun_out/prof_phys -o phys -- python tools/physics_profile.py
    python tools/physics_profile.py --counters      # builds an instrumented copy of libhop.so under tools/_tmp/

The workload is bench.py's `next_rows.n1_physics`: 2048 hypotheses within 8 deg / 4 mm of a grasp of the ellipsoid,
5120-face object mesh, four ~100-face finger meshes, 1964 finger-cloud points, 5000 model points, 20 k scene points.
With --counters the library is compiled with -DSDF_COUNT (hop_sdf.h) and the script prints, per signed-distance query,
the tree nodes visited, faces tested and exact ties resolved.
"""
import argparse
import ctypes as C
import importlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_counting_library():
    src = os.path.join(ROOT, "icra20-hand-object-pose_amd", "csrc")
    out = os.path.join(ROOT, "tools", "_tmp", "cnt")
    os.makedirs(out, exist_ok=True)
    subprocess.check_call(["make", "-C", src], stdout=subprocess.DEVNULL)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
    subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-DSDF_COUNT", "-c", os.path.join(src, "hop_physics.hip"), "-o", os.path.join(out, "hop_physics.o")])
    lib = os.path.join(out, "libhop_count.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", os.path.join(src, "..", "lib", "obj", "hop_kernels.o"),
                           os.path.join(src, "..", "lib", "obj", "hop_ctx.o"), os.path.join(out, "hop_physics.o"), "-o", lib])
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hyps", type=int, default=2048)
    ap.add_argument("--repeat", type=int, default=5)
    ap.add_argument("--counters", action="store_true")
    ap.add_argument("--counting-lib", default=None, help="use an already built instrumented library (GPU box: no compiler run needed)")
    a = ap.parse_args()
    synth = importlib.import_module("icra20-hand-object-pose_amd.synth")
    api = importlib.import_module("icra20-hand-object-pose_amd.api")
    if a.counters:
        api.LIB_PATH = a.counting_lib or build_counting_library()
    elif os.environ.get("HOP_LIB"):
        api.LIB_PATH = os.environ["HOP_LIB"]  # an alternative build of libhop.so (experiments)
    p, poses = synth.physics_case(a.hyps, seed=21, n_model=5000, n_scene=20000, mesh_subdiv=4, spacing=0.003, max_rot_deg=8.0, max_trans=0.004)
    c = api.Context(0)
    for mid, V, F, T in p["meshes"]:
        c.sdf_register_mesh(mid, V, F, T)
    cnt = (C.c_ulonglong * 4)()
    L = api.lib()

    def counters(label, pp):
        c.physics_set_frame(pp)
        c.hypos_upload(poses)
        L.hop_sdf_counters(cnt, 1)
        c.reject_by_collision()
        L.hop_sdf_counters(cnt, 1)
        q = max(cnt[3], 1)
        print(f"{label}: queries {cnt[3]}  nodes/query {cnt[0] / q:.1f}  faces/query {cnt[1] / q:.1f}  ties/query {cnt[2] / q:.2f}")

    if a.counters:
        p_fingers = dict(p)
        p_fingers["model"] = p["model"][:0]
        counters("centre + finger clouds vs object mesh", p_fingers)
        counters("all checks (adds model points vs 4 finger meshes)", p)
        return
    for _ in range(a.repeat):
        c.physics_set_frame(p)
        c.hypos_upload(poses)
        keep, diag = c.reject_by_collision()
        print("set_frame / reject device ms", c.physics_timing(), "kept", int(keep.sum()), "by check", np.bincount(diag[:, 0].astype(int), minlength=6).tolist())


if __name__ == "__main__":
    main()
