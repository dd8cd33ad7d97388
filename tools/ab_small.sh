#!/bin/bash
# usage: ab_small.sh "<variants>" "<shapes>"
cd "${GRAFT_REPO_ROOT:-$PWD}"
mkdir -p gpurun_out/ab
for r in 1 2; do
for v in $1; do
  if [ $v = product ]; then unset ROX_LIB; else export ROX_LIB=$PWD/build/variants/$v.so; fi
  timeout 300 python tools/block_rule_sweep.py --shapes $2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$v', d['shape'], 'full', d.get('full_us'), d.get('full_min_us'), 'hits', d.get('hits_us'), d.get('hits_min_us'))
"
done
done
