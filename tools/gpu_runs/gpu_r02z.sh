#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
(timeout 900 python -m pytest tests/test_gpu_product.py -m gpu -q -x -k "ingested" > gpurun_out/r02z/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02z/pytest.log)
tail -5 gpurun_out/r02z/pytest.log
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02z/bench_driver.json 2> gpurun_out/r02z/bench_driver.err
python -c "
import json; b=json.load(open('gpurun_out/r02z/bench_driver.json')); print(b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline_hits']['kernel_ms'], b['roofline_hits']['valu_issue_frac'], b['spot_diagram']['wallclock_ms']); print(b['psf'])"
tail -3 gpurun_out/r02z/bench_driver.err
(cd /tmp; export TMPDIR=/tmp; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r02z/prof_psf -o psf -- python $GRAFT_REPO_ROOT/tools/psf_bench.py > $GRAFT_REPO_ROOT/gpurun_out/r02z/prof_psf.log 2>&1)
find gpurun_out/r02z/prof_psf -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-200
