mkdir -p gpurun_out/c21
python tools/ab/prio_check.py 3 fdasm_ins820 fdasm_ins822 fdasm_ins824 fdasm_ins825 fdasm_ins826 fdasm_ins828 fdasm_ins830 fdasm_ins832  > gpurun_out/c21/prio.txt 2>&1
cat gpurun_out/c21/prio.txt
