mkdir -p gpurun_out/r02l
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02l/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-strong > $R/gpurun_out/r02l/prof.log 2>&1)
tail -2 gpurun_out/r02l/prof.log | cut -c1-600
bash tools/pmc_collect.sh r02 dblgauss_c2 > gpurun_out/r02l/pmc.log 2>&1
bash tools/pmc_collect.sh r02_cell cell_phone > gpurun_out/r02l/pmc_cell.log 2>&1
bash tools/pmc_collect.sh r02_nikkor nikkor_c3 > gpurun_out/r02l/pmc_nikkor.log 2>&1
tail -2 gpurun_out/r02l/pmc.log
timeout 600 python bench.py > gpurun_out/r02l/bench.json 2> gpurun_out/r02l/bench.err
