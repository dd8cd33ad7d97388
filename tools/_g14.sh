set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05p
mkdir -p gpurun_out/$T
bash tools/gpu_run.sh $T tests smoke
for wl in dblgauss_c2 zmx_evenasph_c3 nikkor_c3 cell_phone; do bash tools/pmc_collect.sh ${T}_$wl $wl > gpurun_out/$T/pmc_$wl.log 2>&1; tail -1 gpurun_out/$T/pmc_$wl.log; done
timeout 300 python tools/figure_latency.py > gpurun_out/$T/figure_latency.jsonl 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/r05p/figure_latency.jsonl'):
    d=json.loads(l)
    if 'RayFan' in d.get('what','') or 'model update' in d.get('what',''): print(d.get('model'), d.get('what','')[:32], 'ref', round(d.get('reference_ms') or 0,2), 'ours', round(d.get('drop_in_ms') or 0,3), d.get('bit_identical'))
PY
timeout 900 python tools/soak_reference.py --hip > gpurun_out/$T/soak_live_reference_gpu.jsonl 2>/dev/null; cut -c1-200 gpurun_out/$T/soak_live_reference_gpu.jsonl
bash tools/gpu_run.sh $T "prof:main:--no-configs --no-strong" prof
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
