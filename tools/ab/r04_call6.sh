mkdir -p gpurun_out/r4c6
O=$PWD/gpurun_out/r4c6
R=$PWD
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl -o tl -- python $R/bench.py --no-cpu-baseline --no-latency --steps 6 --warmup 2 > $O/bench_tl.json 2> $O/bench_tl.err
DB=$(find $O/tl -name "*.db" | head -1); echo DB $DB
python $R/tools/gpu_timeline.py $DB > $O/gpu_timeline.txt 2>&1; head -60 $O/gpu_timeline.txt
timeout 400 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB2 --last-scene > $O/single_scene_kernel_trace.txt 2>&1; head -50 $O/single_scene_kernel_trace.txt
rm -rf $O/tl $O/ss
