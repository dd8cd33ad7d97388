#!/bin/bash
# Round 6's soaks on the GPU box with the final build (gpurun -- 'bash tools/r06_soak_run.sh'):
# fresh seeds for every one of them; records under gpurun_out/r06soak/ (copied to profiles/r06_*).
set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r06soak
mkdir -p $O
timeout 1500 python tools/fuzz_soak.py 700000 30000 > $O/fuzz_soak.json 2> $O/fuzz_soak.err; echo "fuzz rc=$?"
timeout 1500 python tools/fast_soak.py 800000 8000 > $O/fast_soak.json 2> $O/fast_soak.err; echo "fast rc=$?"
ROX_FAST_FP64_FULL=1 timeout 1500 python tools/fast_soak.py 1700000 4000 full > $O/fast_soak_full_packets.json 2> $O/fast_soak_full.err; echo "fast full rc=$?"
timeout 900 python tools/phase_soak.py 200 > $O/phase_soak.json 2> $O/phase_soak.err; echo "phase rc=$?"
timeout 900 python tools/opd_soak.py 60 > $O/opd_soak.json 2> $O/opd_soak.err; echo "opd rc=$?"
timeout 900 python tools/entry_soak.py > $O/entry_soak.json 2> $O/entry_soak.err; echo "entry rc=$?"
timeout 900 python tools/compact_soak.py 300 > $O/compact_and_batch_soak.json 2> $O/compact_soak.err; echo "compact rc=$?"
timeout 1500 python tools/soak_reference.py --hip > $O/soak_live_reference_gpu.jsonl 2> $O/soak_live.err; echo "live rc=$?"
ROX_FORCE_GTAB=1 timeout 900 python tools/fuzz_soak.py 900000 6000 > $O/fuzz_soak_global_table.json 2> $O/fuzz_gtab.err; echo "gtab rc=$?"
for f in $O/*.json; do echo "$f: $(head -c 600 $f)"; done
tail -2 $O/soak_live_reference_gpu.jsonl | cut -c1-300
