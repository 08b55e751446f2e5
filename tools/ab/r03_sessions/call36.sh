mkdir -p gpurun_out/c36
python tools/ab/prio_check.py 4 fdasm_w0 fdasm_pb fdasm_ins581x8 > gpurun_out/c36/prio.txt 2>&1
cat gpurun_out/c36/prio.txt
