R=$PWD; O=$PWD/gpurun_out/r6p1; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB2 --last-scene > $O/single_scene_kernel_trace.txt 2>&1; head -45 $O/single_scene_kernel_trace.txt | cut -c1-150
rm -rf $O/ss
