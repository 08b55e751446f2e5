# steady-state headline under the tracer: per-kernel stats + GPU timeline
R=$PWD; O=$PWD/gpurun_out/r6p2; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -30 $O/bench_kernel_trace.txt | cut -c1-150
python $R/tools/gpu_timeline.py $DB > $O/gpu_timeline.txt 2>&1; cat $O/gpu_timeline.txt | head -40
rm -rf $O/kt
