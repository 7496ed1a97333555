#!/bin/bash
# Profiles of the bench command for profiles/ (run on the GPU box through gpurun):
#   tools/gpu_profile.sh <tag>      -> gpurun_out/<tag>/{stats_default,stats_serial,pmc_fetch,pmc_write,pmc_sq1,pmc_sq2,pmc_tcp}/b_results.db
# Kernel statistics with --kernel-trace --stats only; every counter set in its own pass with --kernel-trace only
# (MI355X_MICROARCH.md: separate --pmc passes).
TAG=${1:-prof}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python /root/repo/bench.py --no-cpu-baseline --no-next-rows --no-alt-modes"
rocprofv3 --kernel-trace --stats -d $OUT/stats_default -o b -- $B --steps 32 --warmup 8 > $OUT/bench_default.json 2> $OUT/bench_default.err
rocprofv3 --kernel-trace --stats -d $OUT/stats_serial -o b -- $B --inflight 1 --steps 8 --warmup 2 > $OUT/bench_serial.json 2> $OUT/bench_serial.err
S="$B --inflight 1 --steps 2 --warmup 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc_fetch -o b -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/pmc_write -o b -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/pmc_sq1 -o b -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/pmc_sq2 -o b -- $S > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc_tcp -o b -- $S > /dev/null 2>&1
find $OUT -name "*.db" | xargs ls -la
# summaries (the databases exceed what gpurun copies back)
cd /root/repo
python tools/rocprof_summary.py $OUT/stats_default/b_results.db $OUT/kernel_stats_default.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-next-rows --steps 32 --warmup 8" > /dev/null
python tools/rocprof_summary.py $OUT/stats_serial/b_results.db $OUT/kernel_stats_serial_inflight1.txt "rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --no-next-rows --inflight 1 --steps 8 --warmup 2" > /dev/null
python tools/pmc_summary.py $OUT/pmc_fetch/b_results.db $OUT/pmc_write/b_results.db $OUT/pmc_latest.json > $OUT/pmc_hbm.txt
python tools/pmc_sq_summary.py $OUT/pmc_sq.txt "rocprofv3 --kernel-trace --pmc <set> -- python bench.py --no-cpu-baseline --no-next-rows --inflight 1 --steps 2 --warmup 1 (three passes: SQ x2, TCP/TCC)" $OUT/pmc_sq1/b_results.db $OUT/pmc_sq2/b_results.db $OUT/pmc_tcp/b_results.db > /dev/null
cp $OUT/pmc_sq.json $OUT/pmc_sq_latest.json 2>/dev/null   # (copy both *_latest.json into profiles/ to have bench.py quote them)
find $OUT -name "*.db" -delete
find $OUT -type d -empty -delete
ls -la $OUT
