#!/usr/bin/env python
"""What can be said about the kernels without a device: per-kernel resource usage and static instruction mix from the gfx950 assembly of HEAD.

    python tools/isa_stats.py [--out profiles/r05_isa_stats.txt]

Compiles csrc/hop_kernels.hip and csrc/hop_icp_lm.hip with hipcc --save-temps -Rpass-analysis=kernel-resource-usage (cross-compiles without a
GPU) and prints, per kernel: VGPRs / AGPRs / SGPRs, scratch, LDS, occupancy (waves per SIMD), instruction counts by class, and for the loops that
hold a marker instruction the count per trip.  Static counts: a divergent wavefront issues every block any of its lanes needs.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "icra20-hand-object-pose_amd", "csrc")
KERNELS = ["k_icp_fusedq_momm", "k_icp_fusedq_momi", "k_icp_fusedq_mom", "k_icp_fusedq", "k_icp_lm7_solve", "k_icp_lm6_solve", "k_lcp_cells_fast", "k_lcp_cells",
           "k_ppf_matrix_sym", "k_pairs", "k_quad_prep", "k_quads", "k_quads_hash", "k_quad_fit", "k_verify_cells", "k_emit", "k_pso_match", "k_pso_outer",
           "k_cell_list_local", "k_topk_merge"]
LOOP_MARKERS = {"k_icp_fusedq_momm": ("v_pk_sub_i16", 8, "packed-list scan, one trip = 2 chunks = 4 candidates"),
                "k_icp_fusedq_momi": ("v_pk_sub_i16", 8, "packed-list scan, one trip = 2 chunks = 4 candidates")}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    lines = ["# gfx950 ISA statistics of HEAD's kernels (hipcc -O3 --offload-arch=gfx950 -ffp-contract=off; tools/isa_stats.py).  NOT measurements.",
             "# kernel | VGPRs AGPRs SGPRs | scratch B/lane | LDS B/block | waves/SIMD | instructions: total VALU SALU VMEM LDS MFMA"]
    with tempfile.TemporaryDirectory() as d:
        for unit in ("hop_kernels", "hop_icp_lm", "hop_comm"):
            # (hop_kernels.hip has flags of its own in csrc/Makefile: packed f32 off as the unit's default, back on per kernel)
            kflags = re.search(r"^KERNELS_FLAGS := (.*)$", open(os.path.join(SRC, "Makefile")).read(), re.M).group(1).split() if unit == "hop_kernels" else []
            r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", *kflags, "-c", os.path.join(SRC, unit + ".hip"),
                                "-o", os.path.join(d, unit + ".o"), "--save-temps", "-Rpass-analysis=kernel-resource-usage"], cwd=d, capture_output=True, text=True)
            if r.returncode != 0:
                sys.exit(r.stderr[-3000:])
            usage = {}
            cur = None
            for ln in r.stderr.splitlines():
                m = re.search(r"Function Name: (\S+)", ln)
                if m:
                    cur = usage.setdefault(m.group(1), {})
                    continue
                m = re.search(r":\s+([A-Za-z][A-Za-z /\[\]]*): (\d+) \[-Rpass", ln)
                if m and cur is not None:
                    cur[m.group(1).strip()] = int(m.group(2))
            asm = open(os.path.join(d, unit + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
            for mangled, u in usage.items():
                short = next((k for k in KERNELS if re.search(r"\d+" + k + r"(E|I)", mangled)), None)
                if short is None:
                    continue
                i = asm.index(mangled + ":")
                body = asm[i:asm.index(".Lfunc_end", i)]
                ops = collections.Counter()
                blocks, curb = [], []
                for ln in body.splitlines():
                    s = ln.strip()
                    if re.match(r"^\.LBB\d+_\d+:", s):
                        blocks.append(curb)
                        curb = []
                        continue
                    m = re.match(r"([a-z_0-9]+)\s", s)
                    if m and not s.startswith(".") and not s.startswith(";"):
                        ops[m.group(1)] += 1
                        curb.append(m.group(1))
                blocks.append(curb)
                cls = lambda p: sum(v for k, v in ops.items() if k.startswith(p))
                mfma = cls("v_mfma")
                tmpl = "" if "I" not in mangled[mangled.index(short) + len(short):][:1] else " (template instance " + mangled[-12:] + ")"
                lines.append(f"{short}{tmpl} | {u.get('VGPRs', 0)} {u.get('AGPRs', 0)} {u.get('TotalSGPRs', 0)} | {u.get('ScratchSize [bytes/lane]', 0)} | "
                             f"{u.get('LDS Size [bytes/block]', 0)} | {u.get('Occupancy [waves/SIMD]', 0)} | {sum(ops.values())} {cls('v_') - mfma} {cls('s_')} "
                             f"{cls('global_') + cls('buffer_') + cls('scratch_') + cls('flat_')} {cls('ds_')} {mfma}")
                if short in LOOP_MARKERS:
                    mark, need, what = LOOP_MARKERS[short]
                    for b in blocks:
                        if sum(1 for x in b if x == mark) >= need:
                            lines.append(f"    loop ({what}): {len(b)} instructions, {sum(1 for x in b if x.startswith('v_'))} VALU, "
                                         f"{sum(1 for x in b if x.startswith('global_'))} VMEM")
                notable = {k: v for k, v in ops.items() if any(t in k for t in ("mfma", "dot2", "perm", "pk_sub", "mad_i32_i16", "med3", "ds_write_b16", "ds_read_b128", "div_scale", "sqrt", "rsq", "rcp"))}
                if notable:
                    lines.append("    " + ", ".join(f"{k} {v}" for k, v in sorted(notable.items())))
    text = "\n".join(lines) + "\n"
    print(text)
    if args.out:
        open(args.out, "w").write(text)


if __name__ == "__main__":
    main()
