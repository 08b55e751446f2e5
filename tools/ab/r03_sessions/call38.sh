mkdir -p gpurun_out/c38
for c in mise128 dense32 stress; do timeout 110 python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/c38/bench_$c.json 2> gpurun_out/c38/bench_$c.err; echo "$c rc $?"; done
for c in mise128 dense32 stress; do cut -c1-200 gpurun_out/c38/bench_$c.json; done
