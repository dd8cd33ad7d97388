#!/bin/bash
# The round-6 evidence run on the GPU box (gpurun -- 'bash tools/r06_profile_run.sh'): the
# driver's bench command, rocprofv3 kernel stats of it (whole and main leg), PMC passes of five
# workloads (FULL / HITS / HITS tolerance mode), and the summaries bench.py and DESIGN.md quote
# -- all taken with the library build of this tree (profiles/*.json carry its source hash).
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r06z
bash tools/gpu_run.sh $T bench prof:main:"--no-configs --no-strong" prof pmc:dblgauss_c2 pmc:litho_c5 pmc:zmx_evenasph_c3 pmc:cell_phone pmc:nikkor_c3 > gpurun_out/${T}_run.log 2>&1
mkdir -p gpurun_out/$T
for w in dblgauss_c2 litho_c5 zmx_evenasph_c3 cell_phone nikkor_c3; do
  python tools/pmc_summary.py gpurun_out/pmc_${T}_$w > gpurun_out/$T/pmc_summary_$w.json
done
python tools/make_valu.py $T gpurun_out/pmc_${T}_dblgauss_c2 gpurun_out/pmc_${T}_litho_c5 gpurun_out/pmc_${T}_zmx_evenasph_c3 gpurun_out/pmc_${T}_cell_phone gpurun_out/pmc_${T}_nikkor_c3 > gpurun_out/$T/make_valu.log 2>&1
cp profiles/valu_per_intersection.json gpurun_out/$T/valu_per_intersection.json
python tools/make_traffic.py gpurun_out/pmc_${T}_dblgauss_c2 gpurun_out/$T/traffic.json "gpurun_out/pmc_${T}_dblgauss_c2 ($T)" > gpurun_out/$T/make_traffic.log 2>&1
# the bench line once more, now that traffic / valu records of THIS build exist
cp gpurun_out/$T/traffic.json profiles/traffic.json
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/$T/bench_with_traffic.json 2> gpurun_out/$T/bench_with_traffic.err
find gpurun_out/$T -name "*kernel_stats.csv" | head
tail -4 gpurun_out/${T}_run.log
