R=$PWD
mkdir -p gpurun_out/c11
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/c11/smoke.txt 2>&1
python -m pytest tests -m gpu -q -p no:cacheprovider > gpurun_out/c11/pytest.txt 2>&1
python bench.py --stats-out gpurun_out/c11/scenes.json > gpurun_out/c11/bench_headline.json 2> gpurun_out/c11/bench_headline.err
python bench.py --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/c11/bench_headline20.json 2> gpurun_out/c11/bench_headline20.err
for c in mise128 dense32 stress; do python bench.py --config $c --steps 3 --warmup 1 > gpurun_out/c11/bench_$c.json 2> gpurun_out/c11/bench_$c.err; done
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c11/kt -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $R/gpurun_out/c11/bench_profiled.json 2> $R/gpurun_out/c11/kt.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c11/kt1 -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 3 --warmup 1 > $R/gpurun_out/c11/bench_single.json 2> $R/gpurun_out/c11/kt1.err
cd $R
python tools/rocpd_stats.py $(find gpurun_out/c11/kt -name "*.db" | head -1) 45 > gpurun_out/c11/kernel_trace.txt 2>&1
python tools/rocpd_stats.py $(find gpurun_out/c11/kt1 -name "*.db" | head -1) --last-scene > gpurun_out/c11/single_scene_kernel_trace.txt 2>&1
python tools/stage_times.py > gpurun_out/c11/stage_times.txt 2>&1
rm -rf gpurun_out/c11/kt gpurun_out/c11/kt1
tail -2 gpurun_out/c11/smoke.txt; tail -3 gpurun_out/c11/pytest.txt; cut -c1-120 gpurun_out/c11/bench_headline.json; cut -c1-120 gpurun_out/c11/bench_headline20.json; grep skip_prop gpurun_out/c11/stage_times.txt
