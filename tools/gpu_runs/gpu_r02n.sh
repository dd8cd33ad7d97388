mkdir -p gpurun_out/r02n
cd $GRAFT_REPO_ROOT
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02n/bench_dist.json 2> gpurun_out/r02n/bench_dist.err
tail -3 gpurun_out/r02n/bench_dist.err
python -c "
import json; d=json.load(open('gpurun_out/r02n/bench_dist.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['spot_diagram']['wallclock_ms'])
print(d['strong_scaling'])"
python - <<'PY'
import json
txt=open('gpurun_out/r02n/bench_dist.json').read()
d=json.loads([l for l in txt.splitlines() if l.startswith('{')][-1])
print('DIST', d['ms_per_step'], d['fence_ms'], d['value'], d['roofline']['frac'])
PY
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-strong > gpurun_out/r02n/bench_20.json 2>/dev/null
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02n/bench_20.json'))
print('N1-20', d['ms_per_step'], d['fence_ms'], d['value'], d['roofline']['frac'], d['spot_diagram']['wallclock_ms'])
PY
