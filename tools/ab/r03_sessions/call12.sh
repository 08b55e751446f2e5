mkdir -p gpurun_out/c12
python tools/ab/prio_check.py 4 fdprio > gpurun_out/c12/prio.txt 2>&1
python tools/ab/prio_check.py 6 fdprio_n1 fdprio_n2 fdprio_n4 fdprio_n8 fdprio_n16 fdprio_n32 >> gpurun_out/c12/prio.txt 2>&1
python tools/ab/prio_check.py 8 fdprio_kb fdprio_ka >> gpurun_out/c12/prio.txt 2>&1
cat gpurun_out/c12/prio.txt
