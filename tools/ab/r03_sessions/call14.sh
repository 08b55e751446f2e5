mkdir -p gpurun_out/c14
python tools/ab/prio_check.py 6 fdasm_a0 fdasm_up fdasm_up1 fdasm_up3 fdasm_bn32 fdasm_bn512 fdasm_tn64 > gpurun_out/c14/prio.txt 2>&1
cat gpurun_out/c14/prio.txt
