#!/bin/bash
# tools/build_variant.sh <name> [extra hipcc flags]: an alternative build of libhop.so under tools/_tmp/<name>/ (A/B runs on one box:
# HOP_LIB=tools/_tmp/<name>/libhop.so python tools/icp_bench.py ...)
set -e
N=$1; shift
D=/root/repo/tools/_tmp/$N
mkdir -p $D
cd /root/repo/icra20-hand-object-pose_amd/csrc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result $*"
KF=$(sed -n 's/^KERNELS_FLAGS := //p' Makefile)   # (the per-unit flags of hop_kernels.hip: packed f32 off as the unit default, csrc/Makefile)
[ -n "$HOP_VARIANT_PACKED_F32" ] && KF=""         # (A/B: k_icp_fusedq_momm and k_lcp_cells_fast with the compiler's packed f32 pairs, as before round 6)
for f in hop_kernels hop_ctx hop_icp_lm; do if [ $f = hop_kernels ]; then X="$KF"; else X=""; fi; /opt/rocm/bin/hipcc $FLAGS $X -c $f.hip -o $D/$f.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/hop_kernels.o $D/hop_ctx.o $D/hop_icp_lm.o ../lib/obj/hop_physics.o ../lib/obj/hop_normals.o ../lib/obj/hop_render.o ../lib/obj/hop_comm.o -ldl -o $D/libhop.so
ls -la $D/libhop.so
