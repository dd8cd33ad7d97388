#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02am
run() { # label env...
  local label=$1; shift
  env "$@" timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full $label', round(d['mean_us'],1))" | tee -a gpurun_out/r02am/variants.txt
}
run "1024 hw-dispatch (shipped)" X=1
run "512 hw-dispatch" ROX_LIB=$PWD/build/variants/full512.so
run "512 persistent 2/CU" ROX_LIB=$PWD/build/variants/full512.so ROX_BLOCKS_PER_CU=2
run "512 persistent 3/CU" ROX_LIB=$PWD/build/variants/full512.so ROX_BLOCKS_PER_CU=3
run "256 persistent 5/CU" ROX_LIB=$PWD/build/variants/full256.so ROX_BLOCKS_PER_CU=5
run "256 hw-dispatch" ROX_LIB=$PWD/build/variants/full256.so
