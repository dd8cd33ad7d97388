set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05d
mkdir -p gpurun_out/$T
WLS="dblgauss_c2 zmx_evenasph_c3 nikkor_c3 cell_phone litho_c5"
bash tools/ab_matrix.sh gpurun_out/$T/ab_check.jsonl "sharded" "zmx_evenasph_c3 dblgauss_c2" --check --num 300
bash tools/ab_matrix.sh gpurun_out/$T/ab.jsonl "product sharded" "$WLS"
for n in 3 4 6; do
  export ROX_TICKET_BLOCKS_PER_CU=$n
  echo "--- ROX_TICKET_BLOCKS_PER_CU=$n"
  bash tools/ab_matrix.sh gpurun_out/$T/ab_bpc$n.jsonl "sharded" "$WLS"
done
unset ROX_TICKET_BLOCKS_PER_CU
ROX_LIB=$PWD/build/variants/sharded.so timeout 300 python tools/block_rule_sweep.py --shapes dg64,dg256,dg3x256,dg512,dg724,dg1024,c4,zmx512 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['lib'], d['waves'], 'full', d['full_us'], 'hits', d['hits_us'])"
