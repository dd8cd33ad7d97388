#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ao
for v in default ident default ident; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  for m in hits full; do
  env $L timeout 100 python tools/sustained_probe.py --mode $m --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m $v', round(d['mean_us'],1))" | tee -a gpurun_out/r02ao/variants.txt
  done
done
ROX_LIB=$PWD/build/variants/ident.so timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
