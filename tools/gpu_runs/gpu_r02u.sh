#!/bin/bash
# single-ray seam: GPU test + latency
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_product.py -x -q -m gpu -k "single_ray" > gpurun_out/u_test.log 2>&1
echo "test rc=$?" >> gpurun_out/u_test.log
timeout 300 python tools/single_ray_latency.py dblgauss_c2 3000 > gpurun_out/u_single.json 2> gpurun_out/u_single.err
timeout 300 python tools/single_ray_latency.py nikkor_c3 2000 >> gpurun_out/u_single.json 2>> gpurun_out/u_single.err
tail -5 gpurun_out/u_test.log; cat gpurun_out/u_single.json; tail -5 gpurun_out/u_single.err
