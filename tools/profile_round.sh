#!/bin/bash
# Round profile of the bench command (run on the GPU box through gpurun, from the repo root):
#   tools/profile_round.sh TAG [WORKLOAD] [COMMIT]
#   1. rocprofv3 --kernel-trace --stats of `python3 bench.py --workload W` (per-kernel durations)
#   2. separate --pmc passes (FETCH_SIZE, WRITE_SIZE cannot share a pass on gfx950; SQ counters in a third)
# Summaries land in gpurun_out/prof_TAG/ ; copy the ones to keep into profiles/TAG/.  COMMIT (the box has no .git) is
# recorded in bench_pmc[_W].meta.json, which bench.py quotes next to roofline.traffic.
set -e
TAG=${1:-r01}
W=${2:-c3}
COMMIT=${3:-unrecorded}
SUF=""; [ "$W" != "c3" ] && SUF="_$W"
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# (--subshards 1: whole-shard launches, so that per-kernel durations and counters describe one launch per kernel and iteration;
#  the default run at the end uses the sub-shard streams)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace$SUF -- python3 $R/bench.py --workload $W --steps 50 --warmup 5 --no-cpu-baseline --subshards 1 > $O/bench_line_under_rocprof$SUF.json 2> $O/trace$SUF.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch$SUF -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_fetch$SUF.log
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write$SUF -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_write$SUF.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq$SUF -- python3 $R/bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 > /dev/null 2> $O/pmc_sq$SUF.log
python3 $R/tools/pmc_summary.py $O/bench_pmc$SUF.csv $O/pmc_fetch$SUF $O/pmc_write$SUF $O/pmc_sq$SUF
echo "{\"commit\": \"$COMMIT\", \"workload\": \"$W\", \"command\": \"bench.py --workload $W --steps 20 --warmup 5 --no-cpu-baseline --subshards 1 under rocprofv3 --pmc (three passes)\"}" > $O/bench_pmc$SUF.meta.json
cp $(find $O/trace$SUF -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats$SUF.csv
python3 $R/bench.py --workload $W --steps 50 --warmup 5 --subshards 1 > $O/bench_line_subshards1$SUF.json 2> $O/bench1$SUF.log
python3 $R/bench.py --workload $W --steps 50 --warmup 5 > $O/bench_line$SUF.json 2> $O/bench$SUF.log
cat $O/bench_line$SUF.json
