# round 5, GPU call 4: FPS abort tests (16 hardware queues, probe), then the MISE / decoder / full-size parity tests
mkdir -p gpurun_out/r5c4
O=$PWD/gpurun_out/r5c4
timeout 300 python -m pytest tests/test_gpu_fps_abort.py -m gpu -q -p no:cacheprovider > $O/pytest_abort.txt 2>&1; tail -25 $O/pytest_abort.txt | cut -c1-220
timeout 900 python -m pytest tests/test_gpu_generator.py tests/test_gpu_fullsize.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt | cut -c1-220
