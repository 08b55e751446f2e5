mkdir -p gpurun_out/c40
( timeout 80 ./rfdnet_amd/lib/micro/lds_return_race 20000 ) > gpurun_out/c40/lds_return_race.txt 2>&1
cat gpurun_out/c40/lds_return_race.txt
