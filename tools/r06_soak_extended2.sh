#!/bin/bash
cd "${GRAFT_REPO_ROOT:-$PWD}"
O=gpurun_out/r06soak_ext2
mkdir -p $O
timeout 1500 python tools/fuzz_soak.py 2000000 60000 > $O/fuzz_soak.json 2> $O/fuzz.err; echo "fuzz rc=$?"
ROX_FORCE_GTAB=1 timeout 900 python tools/fuzz_soak.py 2100000 10000 > $O/fuzz_soak_global_table.json 2> $O/gtab.err; echo "gtab rc=$?"
ROX_FAST_FP64_FULL=1 timeout 900 python tools/fast_soak.py 2200000 6000 full > $O/fast_soak_full_packets.json 2> $O/fastfull.err; echo "fast full rc=$?"
timeout 600 python tools/compact_soak.py 600 > $O/compact_and_batch_soak.json 2> $O/compact.err; echo "compact rc=$?"
for f in $O/*.json; do echo "$f: $(head -c 420 $f)"; done
