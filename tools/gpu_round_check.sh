#!/bin/bash
# First thing to run on an MI355X box when GPU access (re)opens (VERDICT r03 "next round" item 1): everything the driver runs at round
# end, with the logs KEPT under gpurun_out/<tag>/ so that the summaries can be committed to profiles/.
#   gpurun --timeout 3000 -- 'bash tools/gpu_round_check.sh r04'
#     gputest.log      python -m pytest tests -m gpu  (no -x: every red test is listed; durations of the slowest 20)
#     smoke.log        __graft_entry__.smoke()
#     bench.json/.err  python bench.py --gpus 1 --steps 20 --warmup 5   (the driver's command)
#     bench_*.json     the same with nn_mode 6 (round 3's default), with nn_mode 7's sums on the vector units (HOP_ICP_MFMA=0), with k_quads_hash
#     icp_bench_*.json the ICP stage alone: nn_mode 4 / 6 / 7 (matrix cores) / 7dot2 (vector units), and nn_mode 7 built for 5 and 8 waves per SIMD (default 6)
# Profiles (rocprofv3 kernel stats, PMC passes) are a second call: tools/gpu_profile.sh <tag>.
TAG=${1:-r06}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=20 --timeout 900 > $OUT/gputest.log 2>&1
echo "pytest exit $?" >> $OUT/gputest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --nn-mode 6 --no-cpu-baseline --no-next-rows --no-alt-modes > $OUT/bench_nn_mode6.json 2> $OUT/bench_nn_mode6.err
HOP_QUADS_HASH=1 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-alt-modes > $OUT/bench_quads_hash.json 2> $OUT/bench_quads_hash.err
HOP_ICP_MFMA=0 timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-alt-modes > $OUT/bench_nn_mode7_dot2.json 2> $OUT/bench_nn_mode7_dot2.err
# the ICP stage alone (one frame at a time, HIP events): nn_mode 4 / 6 / 7 on the matrix cores / 7 on the vector units, and k_icp_fusedq_momm built for
# 5, 7 and 8 waves per SIMD (default 6 since round 6: 80 VGPRs, no scratch; 5: 87 VGPRs; 7: 72 + 9 dwords of scratch; 8: 64 + 25)
timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 4,6,7,7dot2 --lcp-modes 3 > $OUT/icp_bench_modes_4_6_7.json 2> $OUT/icp_bench.err
# computeLCP's reduced-sum kernel: range records instead of the inline-head records (round 5), and the scene lists at subdivision 3 (shorter lists,
# which the head records make cheaper; more cells to build) -- lcp.3.ms_cells of the three files is the comparison
HOP_LCP_NO_HEAD=1 timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 7 --lcp-modes 3 > $OUT/icp_bench_lcp_range_records.json 2>> $OUT/icp_bench.err
HOP_LCP_TILES=1 timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 7 --lcp-modes 3 > $OUT/icp_bench_lcp_one_tile_per_wave.json 2>> $OUT/icp_bench.err
HOP_LCP_SCENE_SUB=3 timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 7 --lcp-modes 3 > $OUT/icp_bench_lcp_scene_sub3.json 2>> $OUT/icp_bench.err
for W in 5 7 8; do
  bash tools/build_variant.sh momm$W -DHOP_ICP_MOMM_W=$W > $OUT/build_momm$W.log 2>&1 && HOP_LIB=tools/_tmp/momm$W/libhop.so timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 7 --lcp-modes 3 > $OUT/icp_bench_mode7_${W}waves.json 2>> $OUT/icp_bench.err
done
# round 6's scalar-f32 build of the two lookup kernels against the compiler's packed pairs (the only difference of this variant): lcp.3.ms_cells and the
# icp.7 figures of the two files are the comparison
HOP_VARIANT_PACKED_F32=1 bash tools/build_variant.sh pkf32 > $OUT/build_pkf32.log 2>&1 && HOP_LIB=tools/_tmp/pkf32/libhop.so timeout 600 python tools/icp_bench.py --reps 3 --icp-modes 7 --lcp-modes 3 > $OUT/icp_bench_mode7_packed_f32.json 2>> $OUT/icp_bench.err
tail -15 $OUT/gputest.log
tail -2 $OUT/smoke.log
head -c 1500 $OUT/bench.json
echo
python - <<PY
import json
for f in ("bench", "bench_nn_mode6", "bench_nn_mode7_dot2", "bench_quads_hash"):
    try:
        d = json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        print(f, d["value"], d["unit"], d["ms_per_step"], "ms/step", d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
    except Exception as e:
        print(f, "unreadable:", e)
PY
