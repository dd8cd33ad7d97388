#!/bin/bash
# usage: ab_full.sh "<variants>" "<workloads>" "<modes>"   (sustained_probe.py, two alternating rounds)
cd "${GRAFT_REPO_ROOT:-$PWD}"
for r in 1 2; do
for w in $2; do
for m in $3; do
for v in $1; do
  if [ $v = product ]; then unset ROX_LIB; else export ROX_LIB=$PWD/build/variants/$v.so; fi
  timeout 120 python tools/sustained_probe.py --mode $m --workload $w --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m $w $v', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))"
done; done; done; done
