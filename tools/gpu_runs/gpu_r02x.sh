#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02x
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02x/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02x/pytest.log)
tail -4 gpurun_out/r02x/pytest.log
timeout 300 python tools/single_ray_latency.py dblgauss_c2 3000 > gpurun_out/r02x/single.jsonl 2> gpurun_out/r02x/single.err
timeout 300 python tools/single_ray_latency.py nikkor_c3 2000 >> gpurun_out/r02x/single.jsonl 2>> gpurun_out/r02x/single.err
cat gpurun_out/r02x/single.jsonl
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
