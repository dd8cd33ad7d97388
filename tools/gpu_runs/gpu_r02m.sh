mkdir -p gpurun_out/r02m
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_r02.py tests/test_gpu_product.py -m gpu -q -x > gpurun_out/r02m/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02m/pytest.log)
tail -3 gpurun_out/r02m/pytest.log
timeout 200 python tools/spot_wallclock.py --reps 80 2>/dev/null | tee gpurun_out/r02m/spot.json
