#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
(timeout 600 python -m pytest tests/test_psf.py -m gpu -q -x > gpurun_out/r02y/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02y/pytest.log)
tail -15 gpurun_out/r02y/pytest.log
timeout 300 python tools/psf_bench.py > gpurun_out/r02y/psf.jsonl 2> gpurun_out/r02y/psf.err
cat gpurun_out/r02y/psf.jsonl; tail -3 gpurun_out/r02y/psf.err
