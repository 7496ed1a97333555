#!/bin/bash
# The short form of tools/gpu_round_check.sh for a box that is only available for a few minutes: smoke(), the instruction self-test, the driver's
# bench command, the nn_mode 7 bit tests at small sizes -- logs under gpurun_out/<tag>/.
TAG=${1:-r06q}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
echo "smoke exit $?" >> $OUT/smoke.log
timeout 300 python -m pytest tests/test_gpu_zzzz_dev_selftest.py -m gpu -q -p no:cacheprovider --timeout 200 > $OUT/selftest.log 2>&1
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python -m pytest tests/test_gpu_zy_icp_canon.py -m gpu -q -p no:cacheprovider --timeout 300 -k "returns_the_oracles_bits or not_converged or scene_order" > $OUT/icp_canon_small.log 2>&1
HOP_ICP_MFMA=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-alt-modes > $OUT/bench_nn_mode7_dot2.json 2> $OUT/bench_nn_mode7_dot2.err
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/gputest_x.log 2>&1
tail -3 $OUT/smoke.log; tail -3 $OUT/selftest.log; tail -3 $OUT/icp_canon_small.log; tail -3 $OUT/gputest_x.log; head -c 1200 $OUT/bench.json
