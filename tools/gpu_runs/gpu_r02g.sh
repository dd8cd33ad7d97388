mkdir -p gpurun_out/r02g
cd $GRAFT_REPO_ROOT
rocm-smi --showmaxpower --showpower 2>&1 | head -20
timeout 100 python tools/sustained_probe.py --mode full > gpurun_out/r02g/sus_full_sync.json 2>/dev/null; cat gpurun_out/r02g/sus_full_sync.json
ROX_LIB=$PWD/build/variants/nosync.so timeout 100 python tools/sustained_probe.py --mode full > gpurun_out/r02g/sus_full_nosync.json 2>/dev/null; cat gpurun_out/r02g/sus_full_nosync.json
timeout 100 python tools/sustained_probe.py --mode hits > gpurun_out/r02g/sus_hits.json 2>/dev/null; cat gpurun_out/r02g/sus_hits.json
