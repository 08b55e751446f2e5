mkdir -p gpurun_out/c18
python tools/ab/prio_check.py 4 fdasm_pn_top1 fdasm_pn_top2 fdasm_pn_top4 fdasm_pn_top8 fdasm_pn_top16 fdasm_pn_top32 > gpurun_out/c18/prio.txt 2>&1
cat gpurun_out/c18/prio.txt
