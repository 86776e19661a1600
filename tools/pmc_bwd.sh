#!/bin/bash
# SQ instruction counters of the backward sweep alone (tools/time_bwd2.py), on the GPU box: tools/pmc_bwd.sh OUTDIR
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-pmc_bwd}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_sq -- python3 $R/tools/time_bwd2.py > $O/time.txt 2> $O/pmc.log
python3 $R/tools/pmc_summary.py $O/pmc.csv $O/pmc_sq
grep backward $O/pmc.csv
