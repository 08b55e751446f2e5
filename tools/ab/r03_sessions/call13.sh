mkdir -p gpurun_out/c13
python tools/ab/prio_check.py 6 fdprio_n128 fdprio_n512 fdprio_v1 fdprio_v4 > gpurun_out/c13/prio.txt 2>&1
cat gpurun_out/c13/prio.txt
