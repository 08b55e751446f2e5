mkdir -p gpurun_out/c15
python tools/ab/prio_check.py 4 fdasm_lg0 fdasm_vm0 fdasm_dn1 fdasm_mn4 fdasm_pn8 fdasm_noprio > gpurun_out/c15/prio.txt 2>&1
cat gpurun_out/c15/prio.txt
