mkdir -p gpurun_out/c7
python -m pytest tests/test_gpu_chain.py tests/test_gpu_net.py tests/test_gpu_gemm.py -m gpu -q -s -p no:cacheprovider > gpurun_out/c7/pytest.txt 2>&1
python tools/stage_times.py > gpurun_out/c7/stage_chain.txt 2>&1
RFD_NO_CHAIN=1 python tools/stage_times.py > gpurun_out/c7/stage_nochain.txt 2>&1
python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c7/bench_chain.json 2> gpurun_out/c7/bench_chain.err
RFD_NO_CHAIN=1 python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c7/bench_nochain.json 2> gpurun_out/c7/bench_nochain.err
python bench.py --no-cpu-baseline --no-latency --steps 8 > gpurun_out/c7/bench_chain2.json 2> gpurun_out/c7/bench_chain2.err
tail -5 gpurun_out/c7/pytest.txt; grep "skip_propagation\|total gpu" gpurun_out/c7/stage_chain.txt gpurun_out/c7/stage_nochain.txt; for f in chain nochain chain2; do cut -c1-130 gpurun_out/c7/bench_$f.json; done
