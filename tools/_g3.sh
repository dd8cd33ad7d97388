set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05c
mkdir -p gpurun_out/$T
bash tools/gpu_run.sh $T tests
WLS="dblgauss_c2 zmx_evenasph_c3 nikkor_c3 cell_phone litho_c5"
bash tools/ab_matrix.sh gpurun_out/$T/ab.jsonl "notick product" "$WLS"
for n in 1 2 3 4; do
  export ROX_TICKET_BLOCKS_PER_CU=$n
  echo "--- ROX_TICKET_BLOCKS_PER_CU=$n"
  bash tools/ab_matrix.sh gpurun_out/$T/ab_bpc$n.jsonl "product" "$WLS"
done
unset ROX_TICKET_BLOCKS_PER_CU
for v in 0 1; do ROX_SMALL_BLOCKS=$v timeout 300 python tools/block_rule_sweep.py --shapes c1,dg64,dg3x64,dg256,dg3x256,dg512,dg640,dg724,dg1024,c4,zmx512,nik512,phone512 >> gpurun_out/$T/block_rule.jsonl 2>gpurun_out/$T/block_rule.err; done
cat gpurun_out/$T/block_rule.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['small_blocks_env'], d['waves'], 'full', d['full_us'], 'hits', d['hits_us'])"
