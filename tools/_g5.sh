set -u
cd "${GRAFT_REPO_ROOT:-$PWD}"
T=r05e
mkdir -p gpurun_out/$T
timeout 1200 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_configs.py tests/test_gpu_r05.py tests/test_gpu_parity.py tests/test_gpu_r02.py -m gpu -x -q 2>&1 | tail -4
bash tools/ab_matrix.sh gpurun_out/$T/ab_check.jsonl "product" "zmx_evenasph_c3" --check --num 300
echo "--- auto"
bash tools/ab_matrix.sh gpurun_out/$T/ab.jsonl "product" "zmx_evenasph_c3 nikkor_c3"
export ROX_FORCE_INSTANCE=6
echo "--- ROX_FORCE_INSTANCE=6 (general)"
bash tools/ab_matrix.sh gpurun_out/$T/ab_general.jsonl "product" "zmx_evenasph_c3 nikkor_c3"
export ROX_FORCE_INSTANCE=5
echo "--- ROX_FORCE_INSTANCE=5 (evenap)"
bash tools/ab_matrix.sh gpurun_out/$T/ab_evenap.jsonl "product" "nikkor_c3 dblgauss_c2"
export ROX_FORCE_INSTANCE=4
echo "--- ROX_FORCE_INSTANCE=4 (aplist)"
bash tools/ab_matrix.sh gpurun_out/$T/ab_aplist.jsonl "product" "dblgauss_c2"
export ROX_FORCE_INSTANCE=1
echo "--- ROX_FORCE_INSTANCE=1 (even)"
bash tools/ab_matrix.sh gpurun_out/$T/ab_even.jsonl "product" "dblgauss_c2"
unset ROX_FORCE_INSTANCE
