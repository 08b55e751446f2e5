mkdir -p gpurun_out/c23
( timeout 300 ./rfdnet_amd/lib/micro/mfma_valu_war 2000 ) > gpurun_out/c23/mfma_valu_war.txt 2>&1
python tools/ab/prio_check.py 3 fdasm_mv834_836 fdasm_swap832 fdasm_swap833  > gpurun_out/c23/prio.txt 2>&1
cat gpurun_out/c23/prio.txt; grep -c BAD gpurun_out/c23/mfma_valu_war.txt; grep BAD gpurun_out/c23/mfma_valu_war.txt | head -20; tail -1 gpurun_out/c23/mfma_valu_war.txt
