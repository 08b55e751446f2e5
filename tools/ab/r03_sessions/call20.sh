mkdir -p gpurun_out/c20
python tools/ab/prio_check.py 3 fdasm_ins738 fdasm_ins754 fdasm_ins770 fdasm_ins786 fdasm_ins802 fdasm_ins818 fdasm_ins834  > gpurun_out/c20/prio.txt 2>&1
cat gpurun_out/c20/prio.txt
