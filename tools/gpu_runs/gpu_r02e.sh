set -x
mkdir -p gpurun_out/r02e
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/r02e/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02e/pytest.log)
tail -8 gpurun_out/r02e/pytest.log
timeout 900 python bench.py > gpurun_out/r02e/bench.json 2> gpurun_out/r02e/bench.err
cat gpurun_out/r02e/bench.json; tail -3 gpurun_out/r02e/bench.err
timeout 300 python tools/model_table.py > gpurun_out/r02e/models.json 2> gpurun_out/r02e/models.err
# rocprofv3 kernel trace of the bench command (no PMC in this pass)
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02e/prof -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --no-cpu-baseline --no-strong > $GRAFT_REPO_ROOT/gpurun_out/r02e/prof.log 2>&1)
find gpurun_out/r02e/prof -name "*kernel_stats.csv" | head -3
# PMC passes (separate runs)
bash tools/pmc_collect.sh r02 dblgauss_c2 > gpurun_out/r02e/pmc.log 2>&1
bash tools/pmc_collect.sh r02_cell cell_phone > gpurun_out/r02e/pmc_cell.log 2>&1
bash tools/pmc_collect.sh r02_nikkor nikkor_c3 > gpurun_out/r02e/pmc_nikkor.log 2>&1
tail -3 gpurun_out/r02e/pmc.log
