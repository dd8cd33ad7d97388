#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02aj
for v in default unitf default unitf; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  for m in hits full; do
  env $L timeout 100 python tools/sustained_probe.py --mode $m --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m $v', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02aj/variants.txt
  done
done
ROX_LIB=$PWD/build/variants/unitf.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "random_ray_differential or slim or full_size" 2>&1 | tail -2
