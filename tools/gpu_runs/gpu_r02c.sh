set -x
mkdir -p gpurun_out/r02c
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_r02.py -m gpu -q -k "compact or streams or chunked" > gpurun_out/r02c/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02c/pytest.log)
tail -5 gpurun_out/r02c/pytest.log
timeout 300 python tools/spot_wallclock.py > gpurun_out/r02c/spot.json 2> gpurun_out/r02c/spot.err
cat gpurun_out/r02c/spot.json; tail -3 gpurun_out/r02c/spot.err
timeout 300 python tools/spot_wallclock.py --field 2 > gpurun_out/r02c/spot_f2.json 2> gpurun_out/r02c/spot_f2.err
cat gpurun_out/r02c/spot_f2.json
