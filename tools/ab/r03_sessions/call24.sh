mkdir -p gpurun_out/c24
python tools/ab/prio_check.py 3 fdasm_ins826x8 fdasm_ins828x8 fdasm_ins830x8 fdasm_ins832x8 fdasm_ins833x8 fdasm_ins836x8 fdasm_ins838x8 fdasm_ins846x8  > gpurun_out/c24/prio.txt 2>&1
cat gpurun_out/c24/prio.txt
