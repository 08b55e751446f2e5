mkdir -p gpurun_out/c37
python tools/ab/prio_check.py 5 shasm_a0 shasm_ins10x3 shasm_ins10x7 > gpurun_out/c37/prio.txt 2>&1
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/c37/pytest.txt 2>&1
python bench.py > gpurun_out/c37/bench.json 2> gpurun_out/c37/bench.err
python bench.py --no-cpu-baseline --steps 8 --warmup 2 > gpurun_out/c37/bench8.json 2> gpurun_out/c37/bench8.err
grep PRIOCHECK gpurun_out/c37/prio.txt; tail -3 gpurun_out/c37/pytest.txt; cut -c1-220 gpurun_out/c37/bench.json; cut -c1-220 gpurun_out/c37/bench8.json
