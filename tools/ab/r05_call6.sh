mkdir -p gpurun_out/r5c6
O=$PWD/gpurun_out/r5c6
timeout 300 python -m pytest tests/test_gpu_fps_abort.py -m gpu -q -s -p no:cacheprovider > $O/pytest_abort.txt 2>&1; tail -25 $O/pytest_abort.txt | cut -c1-220
