bash tools/gpu_run.sh r06p bench prof:main:"--no-configs --no-strong" prof pmc:dblgauss_c2 pmc:litho_c5 pmc:zmx_evenasph_c3 pmc:cell_phone pmc:nikkor_c3 > gpurun_out/r06p_run.log 2>&1
for w in dblgauss_c2 litho_c5 zmx_evenasph_c3 cell_phone nikkor_c3; do python tools/pmc_summary.py gpurun_out/pmc_r06p_$w > gpurun_out/r06p/pmc_summary_$w.json; done
cp profiles/valu_per_intersection.json gpurun_out/r06p/valu_before.json
python tools/make_valu.py r06p gpurun_out/pmc_r06p_dblgauss_c2 gpurun_out/pmc_r06p_litho_c5 gpurun_out/pmc_r06p_zmx_evenasph_c3 gpurun_out/pmc_r06p_cell_phone gpurun_out/pmc_r06p_nikkor_c3 > gpurun_out/r06p/make_valu.log 2>&1
cp profiles/valu_per_intersection.json gpurun_out/r06p/valu_per_intersection.json
find gpurun_out/r06p -name "*kernel_stats.csv" | head
tail -5 gpurun_out/r06p_run.log
