mkdir -p gpurun_out/c4
( ./rfdnet_amd/lib/micro/mfma_war 1200 1 ) > gpurun_out/c4/mfma_war_bcast.txt 2>&1
python tools/ab/dec_ab.py base r0f r1 p2nt q1nt q2 q2nt > gpurun_out/c4/ab.txt 2>&1
python tools/ab/prio_check.py 10 q2ntp > gpurun_out/c4/prio.txt 2>&1
python -m pytest tests/test_gpu_decoder.py tests/test_gpu_generator.py -m gpu -q -p no:cacheprovider > gpurun_out/c4/pytest.txt 2>&1
tail -2 gpurun_out/c4/pytest.txt; cat gpurun_out/c4/prio.txt; cat gpurun_out/c4/ab.txt; grep -c BAD gpurun_out/c4/mfma_war_bcast.txt; tail -1 gpurun_out/c4/mfma_war_bcast.txt
