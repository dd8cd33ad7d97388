#!/bin/bash
# PMC of the PSF GEMM kernel at (1024, 4096): MFMA busy cycles / instructions
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/pmc_psf
mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
export PSF_ONLY_LARGE=1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F64 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/mfma -o mfma -- python $R/tools/psf_bench.py > $OUT/mfma.log 2>&1
echo rc=$?
python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for p in glob.glob('$OUT/mfma/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(p)):
        if 'cgemm' in r['Kernel_Name']:
            k='cgemm_nt<1>' if 'Li1E' in r['Kernel_Name'] or '<1>' in r['Kernel_Name'] else 'cgemm_nt<0>'
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
import json
print(json.dumps({k:{c:sum(v)/len(v) for c,v in cs.items()} for k,cs in acc.items()}, indent=1))
PY
