mkdir -p gpurun_out/r02p
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_r02.py -m gpu -q -x -k "vignetting or aiming" > gpurun_out/r02p/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02p/pytest.log)
tail -15 gpurun_out/r02p/pytest.log
