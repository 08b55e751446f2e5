mkdir -p gpurun_out/c10
python -m pytest tests/test_gpu_chain.py tests/test_gpu_net.py tests/test_gpu_guards.py -m gpu -q -s -p no:cacheprovider > gpurun_out/c10/pytest.txt 2>&1
python tests/fresh_process_check.py 3 > gpurun_out/c10/fresh3.txt 2>&1
python tools/stage_times.py > gpurun_out/c10/stage_chain.txt 2>&1
RFD_NO_CHAIN=1 python tools/stage_times.py > gpurun_out/c10/stage_nochain.txt 2>&1
python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c10/bench_chain.json 2> gpurun_out/c10/bench_chain.err
RFD_NO_CHAIN=1 python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c10/bench_nochain.json 2> gpurun_out/c10/bench_nochain.err
python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c10/bench_chain2.json 2> gpurun_out/c10/bench_chain2.err
tail -4 gpurun_out/c10/pytest.txt; grep "head B\|log-prob" gpurun_out/c10/pytest.txt; tail -2 gpurun_out/c10/fresh3.txt; grep "skip_propagation" gpurun_out/c10/stage_chain.txt gpurun_out/c10/stage_nochain.txt; for f in chain nochain chain2; do cut -c1-130 gpurun_out/c10/bench_$f.json; done
