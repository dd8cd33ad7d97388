set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05b
mkdir -p gpurun_out/$T
bash tools/gpu_run.sh $T tests
bash tools/ab_matrix.sh gpurun_out/$T/ab.jsonl "notick product" "dblgauss_c2 zmx_evenasph_c3 nikkor_c3 cell_phone litho_c5 rc_telescope_c4"
for v in 0 1; do ROX_SMALL_BLOCKS=$v timeout 300 python tools/block_rule_sweep.py --shapes c1,dg64,dg3x64,dg256,dg3x256,dg512,dg640,dg724,dg1024,c4,zmx512,nik512,phone512 >> gpurun_out/$T/block_rule.jsonl 2>gpurun_out/$T/block_rule.err; done
cat gpurun_out/$T/block_rule.jsonl | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d['shape'], d['small_blocks_env'], d['waves'], 'full', d['full_us'], 'hits', d['hits_us'])"
timeout 200 python tools/single_ray_latency.py dblgauss_c2 2000 > gpurun_out/$T/single_ray.json 2>/dev/null; tail -c 400 gpurun_out/$T/single_ray.json
(cd /tmp; export TMPDIR=/tmp; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/gpurun_out/$T/prof_single -o sr -- python $OLDPWD/tools/single_ray_latency.py dblgauss_c2 500 > /dev/null 2>&1); find gpurun_out/$T/prof_single -name "*kernel_stats.csv" | head -1 | xargs -r head -6 | cut -c1-200
find gpurun_out/$T -name "*kernel_trace.csv" -size +4M -delete
