mkdir -p gpurun_out/c35
( timeout 80 ./rfdnet_amd/lib/micro/lds_return_race 3000 ) > gpurun_out/c35/lds_return_race.txt 2>&1
cat gpurun_out/c35/lds_return_race.txt
