#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02w
(timeout 900 python -m pytest tests/test_gpu_r02.py tests/test_gpu_parity.py -m gpu -q -x -k "host_pointer or pupil_list_entry" > gpurun_out/r02w/pytest.log 2>&1; echo "rc=$?" >> gpurun_out/r02w/pytest.log)
tail -6 gpurun_out/r02w/pytest.log
timeout 300 python tools/host_pointer_latency.py > gpurun_out/r02w/host_pointers.jsonl 2> gpurun_out/r02w/hp.err
cat gpurun_out/r02w/host_pointers.jsonl; tail -3 gpurun_out/r02w/hp.err
