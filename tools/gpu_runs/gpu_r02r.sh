mkdir -p gpurun_out/r02r
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02r/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02r/pytest.log)
tail -4 gpurun_out/r02r/pytest.log
timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))"
timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))"
timeout 300 python tools/model_table.py > gpurun_out/r02r/models.json 2>/dev/null
python -c "
import json; print([(m['model'], m['full_us'], m['hits_us']) for m in json.load(open('gpurun_out/r02r/models.json'))])"
