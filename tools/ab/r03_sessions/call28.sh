mkdir -p gpurun_out/c28
( timeout 150 ./rfdnet_amd/lib/micro/mfma_dst_overlap 300 ) > gpurun_out/c28/mfma_dst_overlap.txt 2>&1
grep -c BAD gpurun_out/c28/mfma_dst_overlap.txt; grep BAD gpurun_out/c28/mfma_dst_overlap.txt | head -40; tail -2 gpurun_out/c28/mfma_dst_overlap.txt; wc -l gpurun_out/c28/mfma_dst_overlap.txt
