mkdir -p gpurun_out/c33
( timeout 90 ./rfdnet_amd/lib/micro/mfma_srcc_raw 100 ) > gpurun_out/c33/mfma_srcc_raw.txt 2>&1
cat gpurun_out/c33/mfma_srcc_raw.txt
