#!/usr/bin/env python
"""tools/cpp_driver_inflight.py -- the measurement behind profiles/r0*_cpp_driver_inflight.txt, reproducible from the tree (VERDICT r03 item 9:
the round-3 numbers came from scripts in git-ignored tools/_tmp).

Writes a synthetic record in the reference's directory layout (rgbN.png / depthN.png 16-bit / palm_in_baseN.txt / arm_left_link_7_t_N.txt;
--distinct rendered grasp frames, each copied --copies times), then runs the C++ dataset driver lib/run_real_all (run_real_all.cpp:70-273
above libhop.so: the whole as-shipped chain from the depth PNG) on one GPU with HOP_FORCE=1 and HOP_INFLIGHT in --inflight, and prints the
driver's own summary lines ("N frames written (A ms per frame alone, K in flight: W ms of wall per frame)") plus, with --timing, its
HOP_APP_TIMING=1 stage table for one frame at a time.

    gpurun -- 'python tools/cpp_driver_inflight.py --distinct 40 --copies 5 --inflight 1 4 8 --timing > gpurun_out/cpp_driver_inflight.txt'
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--distinct", type=int, default=40)
    ap.add_argument("--copies", type=int, default=5)
    ap.add_argument("--inflight", type=int, nargs="+", default=[1, 4, 8])
    ap.add_argument("--timing", action="store_true")
    ap.add_argument("--keep", default="", help="directory to build the dataset in (default: a temporary one)")
    args = ap.parse_args()
    import hop_loader
    hop_loader.load()
    from hop_amd import run_real_all as rr
    lib = os.path.join(ROOT, "icra20-hand-object-pose_amd", "lib")
    exe = os.path.join(lib, "run_real_all")
    work = args.keep or tempfile.mkdtemp(prefix="hop_cppdrv_")
    base = os.path.join(work, "auto_collect")
    rec = rr.write_synthetic_record(base, "ellipse", record="synthetic_000", n_frames=args.distinct)
    cfg_path = os.path.join(base, "config_autodataset.yaml")   # the shipped file with the synthetic camera matrix (write_synthetic_record)
    # copies of the distinct frames under new indices: more frames for a stable mean without more rendering
    names = ("rgb{}.png", "depth{}.png", "palm_in_base{}.txt", "arm_left_link_7_t_{}.txt")
    for c in range(1, args.copies):
        for f in range(args.distinct):
            for n in names:
                src = os.path.join(rec, n.format(f))
                if os.path.exists(src):
                    shutil.copy(src, os.path.join(rec, n.format(c * args.distinct + f)))
    adir = rr.write_assets_dir(rr.Assets(), os.path.join(work, "assets"))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "HOP_GATHER")}
    print(f"# lib/run_real_all on {args.distinct} distinct synthetic grasp frames x {args.copies} copies, HOP_FORCE=1", flush=True)
    for k in args.inflight:
        r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, env=dict(env, HOP_FORCE="1", HOP_INFLIGHT=str(k)))
        line = [ln for ln in r.stdout.splitlines() if "frames written" in ln]
        print(f"{k:2d} in flight: " + (line[-1].split(": ", 1)[1] if line else f"FAILED rc={r.returncode} {r.stderr[-300:]}"), flush=True)
    if args.timing:
        r = subprocess.run([exe, cfg_path, adir, base, "ellipse"], capture_output=True, text=True, env=dict(env, HOP_FORCE="1", HOP_INFLIGHT="1", HOP_APP_TIMING="1"))
        print("## HOP_APP_TIMING=1, one frame at a time (host wall time per stage; asynchronous launches are charged to the stage that waits)")
        for ln in r.stdout.splitlines():
            if ln.startswith("  ") and "ms per frame" in ln:
                print(ln)
    if not args.keep:
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
