#!/bin/bash
# A/B matrix on the GPU box: every variant library under build/variants/ (plus the product
# library as "product") x workloads through tools/ab_bench.py; one JSON line per run.
#   tools/ab_matrix.sh <out.jsonl> "<variants>" "<workloads>" [extra ab_bench args]
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
OUT=$1; VARS=$2; WLS=$3; shift 3
mkdir -p "$(dirname "$OUT")"
for wl in $WLS; do
  for v in $VARS; do
    if [ "$v" = product ]; then unset ROX_LIB; else export ROX_LIB="$PWD/build/variants/$v.so"; fi
    line=$(timeout 300 python tools/ab_bench.py --workload "$wl" --reps 5 --launches 20 "$@" 2>/dev/null | tail -1)
    echo "{\"variant\": \"$v\", \"r\": $line}" >> "$OUT"
    echo "$wl $v $(echo "$line" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', d.get('full_us'), 'hits', d.get('hits_us'), {k:v for k,v in d.items() if k.startswith('bit_exact')})" 2>/dev/null)"
  done
done
unset ROX_LIB
