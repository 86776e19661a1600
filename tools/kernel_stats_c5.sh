#!/bin/bash
# rocprofv3 per-kernel summary of the C5 shard timing script (run on the GPU box through gpurun, from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rp5 -- python3 $R/tools/time_c5.py > $R/gpurun_out/rp5.log 2>&1
cp $(find $R/gpurun_out/rp5 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/c5_kernel_stats.csv
head -9 $R/gpurun_out/c5_kernel_stats.csv | cut -c1-160
