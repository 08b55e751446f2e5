mkdir -p gpurun_out/c2
( ./rfdnet_amd/lib/micro/mfma_war 2000 ) > gpurun_out/c2/mfma_war.txt 2>&1
python tools/ab/prio_check.py 10 fdprio > gpurun_out/c2/prio.txt 2>&1
python tools/ab/prio_check.py 6 fdprio3 >> gpurun_out/c2/prio.txt 2>&1
python tools/ab/prio_check.py 13 r1p >> gpurun_out/c2/prio.txt 2>&1
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/c2/pytest.txt 2>&1
python bench.py --stats-out gpurun_out/c2/scenes.json > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err
python bench.py --config mise128 --steps 2 --warmup 1 > gpurun_out/c2/bench_mise128.json 2> gpurun_out/c2/bench_mise128.err
tail -5 gpurun_out/c2/pytest.txt; cat gpurun_out/c2/prio.txt; grep -c BAD gpurun_out/c2/mfma_war.txt; tail -2 gpurun_out/c2/mfma_war.txt; cut -c1-400 gpurun_out/c2/bench.json; tail -3 gpurun_out/c2/bench.err
