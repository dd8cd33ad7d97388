set -x
mkdir -p gpurun_out/r02b
cd $GRAFT_REPO_ROOT
(timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02b/pytest.log)
tail -5 gpurun_out/r02b/pytest.log
for v in default nosync sync1024 sync256; do
  if [ $v = default ]; then L=""; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 200 python tools/ab_bench.py --check > gpurun_out/r02b/ab_$v.json 2> gpurun_out/r02b/ab_$v.err
  cat gpurun_out/r02b/ab_$v.json
done
timeout 120 ./build/pcie_store > gpurun_out/r02b/pcie_store.jsonl 2>&1
cat gpurun_out/r02b/pcie_store.jsonl
