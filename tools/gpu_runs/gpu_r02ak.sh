#!/bin/bash
# counters + models + bench after the unit() change
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ak
bash tools/pmc_collect.sh r02ak dblgauss_c2 > gpurun_out/r02ak/pmc.log 2>&1
bash tools/pmc_collect.sh r02ak_cell cell_phone > gpurun_out/r02ak/pmc_cell.log 2>&1
bash tools/pmc_collect.sh r02ak_nikkor nikkor_c3 > gpurun_out/r02ak/pmc_nikkor.log 2>&1
timeout 300 python tools/model_table.py > gpurun_out/r02ak/models.json 2>/dev/null
timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 > gpurun_out/r02ak/sustained_hits.json 2>/dev/null
timeout 100 python tools/sustained_probe.py --mode full --seconds 2 > gpurun_out/r02ak/sustained_full.json 2>/dev/null
python -c "
import json
for m in json.load(open('gpurun_out/r02ak/models.json')): print(m['model'], m['full_us'], m['hits_us'])"
