R=$PWD; O=$PWD/gpurun_out/r6p4; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/sd -o sd -- python $R/bench.py --config demo --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 4 --warmup 2 > $O/demo_ss.json 2> $O/demo_ss.err
DB=$(find $O/sd -name "*.db" | head -1); python $R/tools/launch_sequence.py $DB > $O/demo_sequence.txt 2>&1; head -3 $O/demo_sequence.txt
rm -rf $O/sd
