mkdir -p gpurun_out/c26
( timeout 300 ./rfdnet_amd/lib/micro/mfma_srcc_raw 2000 ) > gpurun_out/c26/mfma_srcc_raw.txt 2>&1
cat gpurun_out/c26/mfma_srcc_raw.txt
