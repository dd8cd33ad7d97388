set -x
mkdir -p gpurun_out/r02d
cd $GRAFT_REPO_ROOT
for v in default b256 b128; do
  if [ $v = default ]; then L=""; else L="ROX_LIB=$PWD/build/variants/$v.so"; fi
  env $L timeout 200 python tools/spot_wallclock.py > gpurun_out/r02d/spot_$v.json 2> gpurun_out/r02d/spot_$v.err
  cat gpurun_out/r02d/spot_$v.json
done
