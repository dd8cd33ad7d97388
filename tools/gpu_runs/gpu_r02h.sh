mkdir -p gpurun_out/r02h
cd $GRAFT_REPO_ROOT
run() { # name env...
  name=$1; shift
  env "$@" timeout 100 python tools/sustained_probe.py --mode full --seconds 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', d['lib'], round(d['mean_us'],1), round(d['last_quarter_mean_us'],1))" | tee -a gpurun_out/r02h/full_variants.txt
}
run default X=1
run bpc2 ROX_BLOCKS_PER_CU=2
run bpc3 ROX_BLOCKS_PER_CU=3
run bpc4 ROX_BLOCKS_PER_CU=4
run bpc8 ROX_BLOCKS_PER_CU=8
run sync1024 ROX_LIB=$PWD/build/variants/sync1024.so
run sync1024_bpc2 ROX_LIB=$PWD/build/variants/sync1024.so ROX_BLOCKS_PER_CU=2
run sync256 ROX_LIB=$PWD/build/variants/sync256.so
run nosync ROX_LIB=$PWD/build/variants/nosync.so
