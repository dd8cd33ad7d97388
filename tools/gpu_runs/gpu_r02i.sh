mkdir -p gpurun_out/r02i
cd $GRAFT_REPO_ROOT
for v in default sync1024; do
  if [ $v = default ]; then L="X=1"; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 100 python tools/sustained_probe.py --mode hits --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('hits $v', round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02i/hits_variants.txt
  env $L timeout 200 python tools/spot_wallclock.py --reps 60 2>/dev/null | tee gpurun_out/r02i/spot_$v.json
  env $L timeout 200 python tools/model_table.py 2>/dev/null > gpurun_out/r02i/models_$v.json
done
