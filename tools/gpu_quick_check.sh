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
# one short profile of the same workload, one frame at a time (kernel statistics, then the two HBM counters in their own passes -- MI355X_MICROARCH.md:
# --pmc never together with --stats): what roofline.traffic and profiles/r06_kernel_stats_* need, should this be the only call the round gets
( cd /tmp && export TMPDIR=/tmp
  B="python /root/repo/bench.py --no-cpu-baseline --no-next-rows --no-alt-modes --no-preflight --inflight 1 --steps 4 --warmup 1"
  P=/root/repo/$OUT/prof; mkdir -p $P
  timeout 400 rocprofv3 --kernel-trace --stats -d $P/stats -o b -- $B > $P/bench_serial.json 2> $P/bench_serial.err
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $P/pmc_fetch -o b -- $B > /dev/null 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $P/pmc_write -o b -- $B > /dev/null 2>&1
  cd /root/repo
  python tools/rocprof_summary.py $P/stats/b_results.db $OUT/kernel_stats_serial_inflight1.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-next-rows --no-alt-modes --no-preflight --inflight 1 --steps 4 --warmup 1" > /dev/null 2>> $OUT/prof.err
  python tools/pmc_summary.py $P/pmc_fetch/b_results.db $P/pmc_write/b_results.db $OUT/pmc_latest.json > $OUT/pmc_hbm.txt 2>> $OUT/prof.err
  find $P -name "*.db" -delete; find $P -type d -empty -delete )
HOP_ICP_MFMA=0 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-next-rows --no-alt-modes > $OUT/bench_nn_mode7_dot2.json 2> $OUT/bench_nn_mode7_dot2.err
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --timeout 600 -x > $OUT/gputest_x.log 2>&1
head -12 $OUT/kernel_stats_serial_inflight1.txt 2>/dev/null; tail -3 $OUT/smoke.log; tail -3 $OUT/selftest.log; tail -3 $OUT/icp_canon_small.log; tail -3 $OUT/gputest_x.log; head -c 1200 $OUT/bench.json
