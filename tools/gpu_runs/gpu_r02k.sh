mkdir -p gpurun_out/r02k
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -m gpu -q -x > gpurun_out/r02k/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02k/pytest.log)
tail -3 gpurun_out/r02k/pytest.log
timeout 300 python tools/model_table.py > gpurun_out/r02k/models.json 2>/dev/null
python -c "
import json; print([(m['model'], m['full_us'], m['hits_us']) for m in json.load(open('gpurun_out/r02k/models.json'))])"
