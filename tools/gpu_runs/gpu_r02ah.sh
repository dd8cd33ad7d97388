#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ah
for v in default nt0 default nt0; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full $v', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02ah/variants.txt
done
