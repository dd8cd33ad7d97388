mkdir -p gpurun_out/r02q
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_product.py -m gpu -q -x -k "ingested" > gpurun_out/r02q/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02q/pytest.log)
tail -5 gpurun_out/r02q/pytest.log
for v in default hits256; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits $v', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02q/hits_variants.txt
done
