#!/bin/bash
# final round-2 verification + profile of the bench command
mkdir -p gpurun_out/r02af
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02af/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02af/pytest.log)
tail -3 gpurun_out/r02af/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
(cd /tmp; export TMPDIR=/tmp; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r02af/prof -o bench -- python $R/bench.py --no-cpu-baseline --no-strong > $R/gpurun_out/r02af/prof.log 2>&1)
timeout 600 python bench.py > gpurun_out/r02af/bench.json 2> gpurun_out/r02af/bench.err
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02af/bench_driver.json 2> gpurun_out/r02af/bench_driver.err
python -c "
import json
for f in ('bench','bench_driver'):
    b=json.load(open('gpurun_out/r02af/%s.json'%f)); print(f, b['value'], b['ms_per_step'], b['roofline']['frac'], b['roofline']['kernel_ms'], b['roofline_hits']['kernel_ms'], b['roofline_hits']['valu_issue_frac'], b['spot_diagram']['wallclock_ms'], b['strong_scaling']['end_to_end_ms'], (b['cpu_baseline'] or {}).get('value'))"
head -4 $(find gpurun_out/r02af/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
