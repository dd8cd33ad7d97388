set -x
mkdir -p gpurun_out/r02f
cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_product.py -m gpu -q -k "soa" > gpurun_out/r02f/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02f/pytest.log)
tail -3 gpurun_out/r02f/pytest.log
for rep in 1 2; do
for v in default nosync; do
  if [ $v = default ]; then L=""; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 200 python tools/ab_bench.py --launches 50 --reps 7 > gpurun_out/r02f/ab_${v}_$rep.json 2>/dev/null
  cat gpurun_out/r02f/ab_${v}_$rep.json
  env $L timeout 300 python bench.py --no-strong --no-cpu-baseline --steps 50 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench $v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['spot_diagram']['wallclock_ms'])" | tee -a gpurun_out/r02f/bench_cmp.txt
done
done
