mkdir -p gpurun_out/c3
( ./rfdnet_amd/lib/micro/mfma_war 1500 ) > gpurun_out/c3/mfma_war.txt 2>&1
python tools/ab/prio_check.py 8 fdprio_sb fdprio_sb2 > gpurun_out/c3/prio.txt 2>&1
python tools/ab/prio_check.py 5 fdprio >> gpurun_out/c3/prio.txt 2>&1
python tools/ab/prio_check.py 12 p1p >> gpurun_out/c3/prio.txt 2>&1
python tools/ab/dec_ab.py base r0f r1 p1 p1nt p2 p2nt > gpurun_out/c3/ab.txt 2>&1
python -m pytest tests/test_gpu_guards.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py tests/test_gpu_predictions.py -m gpu -q -s -p no:cacheprovider > gpurun_out/c3/pytest_a.txt 2>&1
python -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -p no:cacheprovider -k "parity or shapes" > gpurun_out/c3/pytest_b.txt 2>&1
python bench.py --no-cpu-baseline --steps 6 > gpurun_out/c3/bench_base.json 2> gpurun_out/c3/bench_base.err
RFD_HIP_LIB=$PWD/rfdnet_amd/lib/variants/librfd_r0f.so python bench.py --no-cpu-baseline --steps 6 > gpurun_out/c3/bench_r0f.json 2> gpurun_out/c3/bench_r0f.err
python bench.py --no-cpu-baseline --steps 6 > gpurun_out/c3/bench_base2.json 2> gpurun_out/c3/bench_base2.err
tail -3 gpurun_out/c3/pytest_a.txt; tail -3 gpurun_out/c3/pytest_b.txt; cat gpurun_out/c3/prio.txt; cat gpurun_out/c3/ab.txt; grep -c BAD gpurun_out/c3/mfma_war.txt; tail -1 gpurun_out/c3/mfma_war.txt; for f in base r0f base2; do cut -c1-160 gpurun_out/c3/bench_$f.json; done
