mkdir -p gpurun_out/c39
RFD_DBG_DUMP=$PWD/gpurun_out/c39/dump python tools/ab/prio_check.py 5 r0old_ins10x7 > gpurun_out/c39/prio.txt 2>&1
cat gpurun_out/c39/prio.txt | head -8; ls gpurun_out/c39/dump | wc -l
