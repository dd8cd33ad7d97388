mkdir -p gpurun_out/r02j
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r02j/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02j/pytest.log)
tail -4 gpurun_out/r02j/pytest.log
timeout 900 python bench.py > gpurun_out/r02j/bench.json 2> gpurun_out/r02j/bench.err
python -c "
import json; d=json.load(open('gpurun_out/r02j/bench.json'))
print('value', d['value'], 'ms/step', d['ms_per_step'], 'roofline', d['roofline']['kernel_ms'], d['roofline']['frac'])
print('hits', d['roofline_hits']['kernel_ms'], d['roofline_hits']['frac'], 'spot', d['spot_diagram']['wallclock_ms'], d['spot_diagram']['wallclock_min_ms'])
print('strong', d['strong_scaling'])
"
timeout 300 python tools/model_table.py > gpurun_out/r02j/models.json 2>/dev/null
python -c "
import json; print([(m['model'], m['full_us'], m['hits_us']) for m in json.load(open('gpurun_out/r02j/models.json'))])"
