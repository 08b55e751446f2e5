mkdir -p gpurun_out/c17
python tools/ab/prio_check.py 4 fdasm_pn_top12 fdasm_dn1_0-4 fdasm_dn1_4-8 fdasm_dn1_8-12 fdasm_dn1_11-12 fdasm_dn1_0-11 > gpurun_out/c17/prio.txt 2>&1
cat gpurun_out/c17/prio.txt
