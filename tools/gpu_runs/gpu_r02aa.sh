#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aa
for b in 0 1 2 4; do
  ROX_BLOCKS_PER_CU=$b timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full blocks_per_cu=$b', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02aa/full_bpc.txt
done
for b in 0 1 2 4; do
  ROX_BLOCKS_PER_CU=$b timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits blocks_per_cu=$b', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02aa/full_bpc.txt
done
