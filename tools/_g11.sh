set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05k
mkdir -p gpurun_out/$T
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_batch.py tests/test_gpu_configs.py tests/test_gpu_timed_launch.py tests/test_gpu_r05.py -m gpu -x -q 2>&1 | tail -3
bash tools/ab_matrix.sh gpurun_out/$T/ab.jsonl "norcp product norcp product" "dblgauss_c2 litho_c5 zmx_evenasph_c3 nikkor_c3"
timeout 300 python tools/block_rule_sweep.py --shapes c1,dg3x64,dg3x256,c4 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['small_blocks_env'], d['waves'], 'full', d['full_us'], 'hits', d['hits_us'])"
