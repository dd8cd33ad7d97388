#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02an
for v in default s_max-ilp s_iterative-maxocc default s_max-ilp s_iterative-maxocc; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  for m in hits full; do
  env $L timeout 100 python tools/sustained_probe.py --mode $m --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m $v', round(d['mean_us'],1))" | tee -a gpurun_out/r02an/variants.txt
  done
done
