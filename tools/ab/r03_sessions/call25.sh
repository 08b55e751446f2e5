mkdir -p gpurun_out/c25
python tools/ab/prio_check.py 3 fdasm_ins821x8 fdasm_ins822x8 fdasm_ins823x8 fdasm_ins825x8 fdasm_ins819x8 fdasm_ins817x8  > gpurun_out/c25/prio.txt 2>&1
cat gpurun_out/c25/prio.txt
