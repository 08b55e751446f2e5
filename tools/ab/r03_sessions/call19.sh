mkdir -p gpurun_out/c19
python tools/ab/prio_check.py 3 fdasm_ins217 fdasm_ins343 fdasm_ins470 fdasm_ins596 fdasm_ins722 fdasm_ins849 fdasm_ins975 fdasm_ins1101 fdasm_ins1228  > gpurun_out/c19/prio.txt 2>&1
cat gpurun_out/c19/prio.txt
