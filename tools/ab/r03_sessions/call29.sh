mkdir -p gpurun_out/c29
python tools/ab/prio_check.py 3 fdasm_ins677x8 fdasm_ins700x8 fdasm_ins760x8 fdasm_ins800x8 > gpurun_out/c29/prio.txt 2>&1
python tools/ab/prio_check.py 6 shasm_a0 shasm_ins10 shasm_ins10x2 shasm_ins10x3 shasm_ins10x4 shasm_ins10x5 shasm_ins10x6 shasm_ins10x7 >> gpurun_out/c29/prio.txt 2>&1
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/c29/pytest.txt 2>&1
python bench.py > gpurun_out/c29/bench.json 2> gpurun_out/c29/bench.err
grep PRIOCHECK gpurun_out/c29/prio.txt; tail -3 gpurun_out/c29/pytest.txt; cut -c1-400 gpurun_out/c29/bench.json
