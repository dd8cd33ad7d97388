mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02n/bench_dist.json 2> gpurun_out/r02n/bench_dist.err
tail -3 gpurun_out/r02n/bench_dist.err
python -c "
import json; d=json.load(open('gpurun_out/r02n/bench_dist.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['spot_diagram']['wallclock_ms'])
print(d['strong_scaling'])"
