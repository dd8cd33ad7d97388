#!/bin/bash
# rocprofv3 kernel stats of the per-model table (all five workloads, FULL and HITS instances)
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/r02ai
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02ai/prof -o models -- python $R/tools/model_table.py > $R/gpurun_out/r02ai/models.json 2> $R/gpurun_out/r02ai/models.err
cut -c1-150 $(find $R/gpurun_out/r02ai/prof -name "*kernel_stats.csv" | head -1) | head -14
