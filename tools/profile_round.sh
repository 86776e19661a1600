#!/bin/bash
# Round profile of the bench command (run on the GPU box through gpurun, from the repo root):
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py` (per-kernel durations)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950; SQ counters in a third)
# Summaries land in gpurun_out/prof_$1/ ; copy the ones to keep into profiles/$1/.
set -e
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (--subshards 1: whole-shard launches, so that per-kernel durations and counters describe one launch per kernel and iteration;
#  the default run at the end uses the sub-shard streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 50 --warmup 5 --no-cpu-baseline --subshards 1 > $O/bench_line_under_rocprof.json 2> $O/trace.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_sq.log
python3 $R/tools/pmc_summary.py $O/bench_pmc.csv $O/pmc_fetch $O/pmc_write $O/pmc_sq
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
python3 $R/bench.py --steps 50 --warmup 5 --subshards 1 > $O/bench_line_subshards1.json 2> $O/bench1.log
python3 $R/bench.py --steps 50 --warmup 5 > $O/bench_line.json 2> $O/bench.log
cat $O/bench_line.json
