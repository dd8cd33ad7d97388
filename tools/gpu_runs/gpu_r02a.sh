set -x
mkdir -p gpurun_out/r02a
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02a/pytest.log)
tail -5 gpurun_out/r02a/pytest.log
timeout 300 python tools/ab_bench.py --check > gpurun_out/r02a/ab_default.json 2> gpurun_out/r02a/ab_default.err
cat gpurun_out/r02a/ab_default.json
ROX_LIB=$PWD/build/variants/wgsync.so timeout 120 python tools/ab_bench.py --check > gpurun_out/r02a/ab_wgsync.json 2> gpurun_out/r02a/ab_wgsync.err
cat gpurun_out/r02a/ab_wgsync.json
timeout 120 ./build/store_desync > gpurun_out/r02a/store_desync.jsonl 2>&1
cat gpurun_out/r02a/store_desync.jsonl
timeout 300 python tools/spot_wallclock.py > gpurun_out/r02a/spot.json 2> gpurun_out/r02a/spot.err
cat gpurun_out/r02a/spot.json; tail -3 gpurun_out/r02a/spot.err
timeout 300 python tools/model_table.py > gpurun_out/r02a/models.json 2> gpurun_out/r02a/models.err
cat gpurun_out/r02a/models.json | head -50
timeout 600 python bench.py --steps 20 > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err
cat gpurun_out/r02a/bench.json
