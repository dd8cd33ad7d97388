#!/bin/bash
# N = 2 rehearsal of bench.py on one GPU (gloo carries the collectives, both ranks on device 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ad
ROX_BENCH_BACKEND=gloo ROX_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r02ad/bench_n2.json 2> gpurun_out/r02ad/bench_n2.err
echo "rc=$?"
python -c "
import json; b=json.load(open('gpurun_out/r02ad/bench_n2.json')); print(b['n_gpus'], b['value'], b['ms_per_step'], b['config']['sharding'], b['config']['rays_per_step']); print(b['strong_scaling'])"
tail -4 gpurun_out/r02ad/bench_n2.err
ROX_BENCH_BACKEND=gloo ROX_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 4 --steps 20 --warmup 5 --strong-num 1024 > gpurun_out/r02ad/bench_n4.json 2> gpurun_out/r02ad/bench_n4.err
echo "rc=$?"
python -c "
import json; b=json.load(open('gpurun_out/r02ad/bench_n4.json')); print(b['n_gpus'], b['value'], b['ms_per_step']); print(b['strong_scaling'])"
