mkdir -p gpurun_out/r02o
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02o/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02o/pytest.log)
tail -4 gpurun_out/r02o/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
