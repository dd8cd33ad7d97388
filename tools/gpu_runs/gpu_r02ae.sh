#!/bin/bash
# N = 2 / 4 rehearsal of bench.py on one GPU (gloo carries the collectives, all ranks on device 0)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ae
for n in 2 4; do
ROX_BENCH_BACKEND=gloo ROX_BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 20 --warmup 5 --strong-num 1024 > gpurun_out/r02ae/bench_n$n.json 2> gpurun_out/r02ae/bench_n$n.err
echo "n=$n rc=$? stdout_lines=$(wc -l < gpurun_out/r02ae/bench_n$n.json) stdout_bytes=$(wc -c < gpurun_out/r02ae/bench_n$n.json)"
python -c "
import json; b=json.load(open('gpurun_out/r02ae/bench_n$n.json')); print(b['n_gpus'], b['value'], b['ms_per_step'], b['config']['rays_per_step']); print({k:b['strong_scaling'][k] for k in ('ranks','backend','rays','kernel_ms_max_over_ranks','gather_ms','end_to_end_ms')})"
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-strong > gpurun_out/r02ae/bench_fd.json 2> gpurun_out/r02ae/bench_fd.err; echo "force-dist rc=$? lines=$(wc -l < gpurun_out/r02ae/bench_fd.json)"; head -c 120 gpurun_out/r02ae/bench_fd.json; echo
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-strong > gpurun_out/r02ae/bench_1.json 2> gpurun_out/r02ae/bench_1.err; echo "n=1 rc=$? lines=$(wc -l < gpurun_out/r02ae/bench_1.json)"; head -c 120 gpurun_out/r02ae/bench_1.json; echo
