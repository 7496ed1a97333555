#!/bin/bash
# The shortest useful call (about ten minutes): smoke(), the instruction self-test, the driver's bench command, one rocprofv3 kernel-statistics pass.
TAG=${1:-r06m}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 240 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log
timeout 200 python -m pytest tests/test_gpu_zzzz_dev_selftest.py -m gpu -q -p no:cacheprovider --timeout 150 > $OUT/selftest.log 2>&1
timeout 420 python bench.py --gpus 1 --steps 20 --warmup 5 --no-next-rows > $OUT/bench.json 2> $OUT/bench.err
( cd /tmp && export TMPDIR=/tmp; P=/root/repo/$OUT/prof; mkdir -p $P
  timeout 300 rocprofv3 --kernel-trace --stats -d $P/stats -o b -- python /root/repo/bench.py --no-cpu-baseline --no-next-rows --no-alt-modes --no-preflight --inflight 1 --steps 4 --warmup 1 > $P/bench_serial.json 2> $P/bench_serial.err
  cd /root/repo; python tools/rocprof_summary.py $P/stats/b_results.db $OUT/kernel_stats_serial_inflight1.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-next-rows --no-alt-modes --no-preflight --inflight 1 --steps 4 --warmup 1" > /dev/null 2>> $OUT/prof.err
  find $P -name "*.db" -delete; find $P -type d -empty -delete )
head -12 $OUT/kernel_stats_serial_inflight1.txt 2>/dev/null; tail -3 $OUT/smoke.log; tail -2 $OUT/selftest.log; head -c 1000 $OUT/bench.json
