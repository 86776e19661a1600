cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/rp -- python3 $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/rp.json 2> $GRAFT_REPO_ROOT/gpurun_out/rp.log
head -7 $(find $GRAFT_REPO_ROOT/gpurun_out/rp -name "*kernel_stats.csv" | head -1) | cut -c1-150
