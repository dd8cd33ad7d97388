set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05m
mkdir -p gpurun_out/$T
bash tools/gpu_run.sh $T tests smoke
bash tools/gpu_run.sh $T bench "prof:main:--no-configs --no-strong" prof
for wl in dblgauss_c2 zmx_evenasph_c3 nikkor_c3 cell_phone; do bash tools/pmc_collect.sh ${T}_$wl $wl > gpurun_out/$T/pmc_$wl.log 2>&1; tail -1 gpurun_out/$T/pmc_$wl.log; done
for v in 0 1; do ROX_SMALL_BLOCKS=$v timeout 300 python tools/block_rule_sweep.py --shapes c1,dg64,dg3x64,dg256,dg3x256,dg512,dg640,dg724,dg1024,c4,zmx512,nik512,phone512 >> gpurun_out/$T/block_rule.jsonl 2>/dev/null; done
timeout 300 python tools/block_rule_sweep.py --shapes c1,dg64,dg3x64,dg256,dg3x256,dg512,dg640,dg724,dg1024,c4,zmx512,nik512,phone512 >> gpurun_out/$T/block_rule.jsonl 2>/dev/null
timeout 300 python tools/figure_latency.py > gpurun_out/$T/figure_latency.jsonl 2>/dev/null; wc -l gpurun_out/$T/figure_latency.jsonl
find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
