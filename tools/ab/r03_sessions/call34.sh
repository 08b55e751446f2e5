mkdir -p gpurun_out/c34
RFD_DBG_DUMP=$PWD/gpurun_out/c34/dump python tools/ab/prio_check.py 8 fdprio > gpurun_out/c34/prio.txt 2>&1
cat gpurun_out/c34/prio.txt | head -12; ls gpurun_out/c34/dump | wc -l
