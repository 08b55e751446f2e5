mkdir -p gpurun_out/c27
python tools/ab/prio_check.py 3 r0asm_a0 r0asm_ins10 r0asm_ins10x2 r0asm_ins10x3 r0asm_ins10x4 r0asm_ins10x5 r0asm_ins10x6 r0asm_ins10x7 > gpurun_out/c27/prio.txt 2>&1
python tools/ab/prio_check.py 3 shasm_a0 shasm_ins10 shasm_ins10x2 shasm_ins10x3 shasm_ins10x4 shasm_ins10x5 shasm_ins10x6 shasm_ins10x7 >> gpurun_out/c27/prio.txt 2>&1
cat gpurun_out/c27/prio.txt
