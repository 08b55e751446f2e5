R=$PWD
mkdir -p gpurun_out/c8
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/c8/pytest.txt 2>&1
python bench.py --stats-out gpurun_out/c8/scenes.json > gpurun_out/c8/bench_headline.json 2> gpurun_out/c8/bench_headline.err
for c in mise128 dense32 stress; do python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/c8/bench_$c.json 2> gpurun_out/c8/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c8/kt -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $R/gpurun_out/c8/bench_profiled.json 2> $R/gpurun_out/c8/kt.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c8/kt1 -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 3 --warmup 1 > $R/gpurun_out/c8/bench_single.json 2> $R/gpurun_out/c8/kt1.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c8/pmc/fetch -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/c8/pmc_fetch.json 2> $R/gpurun_out/c8/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c8/pmc/write -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/c8/pmc_write.json 2> $R/gpurun_out/c8/pmc_write.err
cd $R
python tools/rocpd_stats.py $(find gpurun_out/c8/kt -name "*.db" | head -1) 45 > gpurun_out/c8/kernel_trace.txt 2>&1
python tools/rocpd_stats.py $(find gpurun_out/c8/kt1 -name "*.db" | head -1) --last-scene > gpurun_out/c8/single_scene_kernel_trace.txt 2>&1
python tools/stage_times.py > gpurun_out/c8/stage_times.txt 2>&1
rm -rf gpurun_out/c8/kt gpurun_out/c8/kt1
tail -3 gpurun_out/c8/pytest.txt; cut -c1-120 gpurun_out/c8/bench_headline.json; head -6 gpurun_out/c8/kernel_trace.txt | cut -c1-160; head -30 gpurun_out/c8/single_scene_kernel_trace.txt | cut -c1-150
