#!/bin/bash
# More seeds for the round-6 soaks (gpurun -- 'bash tools/r06_soak_extended.sh'): seeds no earlier
# soak of any round used; records under gpurun_out/r06soak_ext/.
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r06soak_ext
mkdir -p $O
timeout 2400 python tools/fuzz_soak.py 1200000 100000 > $O/fuzz_soak.json 2> $O/fuzz_soak.err; echo "fuzz rc=$?"
ROX_FORCE_GTAB=1 timeout 1200 python tools/fuzz_soak.py 1400000 20000 > $O/fuzz_soak_global_table.json 2> $O/fuzz_gtab.err; echo "gtab rc=$?"
timeout 1500 python tools/fast_soak.py 1500000 16000 > $O/fast_soak.json 2> $O/fast_soak.err; echo "fast rc=$?"
timeout 900 python tools/phase_soak.py 600 > $O/phase_soak.json 2> $O/phase_soak.err; echo "phase rc=$?"
timeout 900 python tools/compact_soak.py 900 > $O/compact_and_batch_soak.json 2> $O/compact_soak.err; echo "compact rc=$?"
for f in $O/*.json; do echo "$f: $(head -c 400 $f)"; done
