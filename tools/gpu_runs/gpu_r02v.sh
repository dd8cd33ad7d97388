#!/bin/bash
# profile refresh of the final round-2 build
mkdir -p gpurun_out/r02v
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02v/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-strong > $R/gpurun_out/r02v/prof.log 2>&1)
tail -2 gpurun_out/r02v/prof.log | cut -c1-400
bash tools/pmc_collect.sh r02v dblgauss_c2 > gpurun_out/r02v/pmc.log 2>&1
bash tools/pmc_collect.sh r02v_cell cell_phone > gpurun_out/r02v/pmc_cell.log 2>&1
bash tools/pmc_collect.sh r02v_nikkor nikkor_c3 > gpurun_out/r02v/pmc_nikkor.log 2>&1
tail -2 gpurun_out/r02v/pmc.log
timeout 600 python bench.py > gpurun_out/r02v/bench.json 2> gpurun_out/r02v/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02v/bench_driver.json 2> gpurun_out/r02v/bench_driver.err
timeout 300 python tools/model_table.py > gpurun_out/r02v/models.json 2>/dev/null
timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 > gpurun_out/r02v/sustained_hits.json 2>/dev/null
timeout 100 python tools/sustained_probe.py --mode full --seconds 2 > gpurun_out/r02v/sustained_full.json 2>/dev/null
cut -c1-1500 gpurun_out/r02v/bench_driver.json
