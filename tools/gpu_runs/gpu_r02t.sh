mkdir -p gpurun_out/r02t
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02t/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02t/pytest.log)
tail -4 gpurun_out/r02t/pytest.log
