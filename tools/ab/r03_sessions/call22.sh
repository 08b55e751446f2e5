mkdir -p gpurun_out/c22
python tools/ab/prio_check.py 3 fdasm_swap823 fdasm_swap824 fdasm_ins824x8 fdasm_ins824x16 fdasm_ins825x7 fdasm_ins824x2  > gpurun_out/c22/prio.txt 2>&1
cat gpurun_out/c22/prio.txt
