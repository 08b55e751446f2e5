mkdir -p gpurun_out/c16
python tools/ab/prio_check.py 4 fdasm_dn1_0-12 fdasm_dn1_12-16 fdasm_dn1_16-20 fdasm_dn1_20-28 fdasm_dn1_12-13 fdasm_dn1_13-14 fdasm_dn1_14-15 fdasm_dn1_15-16 > gpurun_out/c16/prio.txt 2>&1
cat gpurun_out/c16/prio.txt
