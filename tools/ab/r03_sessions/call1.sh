mkdir -p gpurun_out/c1
( ./rfdnet_amd/lib/micro/mfma_war 3000 ) > gpurun_out/c1/mfma_war.txt 2>&1
python tools/ab/prio_check.py 8 r0pn > gpurun_out/c1/prio.txt 2>&1
python tools/ab/prio_check.py 12 r1p r2p >> gpurun_out/c1/prio.txt 2>&1
python tools/ab/dec_ab.py base r0f r1 r2 r2nt r2nosb r0nt r2p > gpurun_out/c1/ab.txt 2>&1
python -m pytest tests/test_gpu_decoder.py -m gpu -x -q > gpurun_out/c1/pytest_dec.txt 2>&1
tail -3 gpurun_out/c1/pytest_dec.txt; cat gpurun_out/c1/prio.txt; cat gpurun_out/c1/ab.txt; tail -5 gpurun_out/c1/mfma_war.txt
