R=$PWD; O=$PWD/gpurun_out/r6p5; mkdir -p $O
for i in 1 2; do python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('headline value %.3f frac %.4f single %.2f' % (d['value'], d['roofline']['frac'], d['single_scene']['ms_per_scene']), d['single_scene']['stage_ms'])"; done
python bench.py --config demo --steps 40 --warmup 5 --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('demo value %.2f single %.2f cpu/scene %.4f' % (d['value'], d['single_scene']['ms_per_scene'], d['config']['host_cpu_s_per_scene_rank0']), d['single_scene']['stage_ms'])"
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/sd -o sd -- python $R/bench.py --config demo --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 4 --warmup 2 > $O/demo_ss.json 2> $O/demo_ss.err
DB=$(find $O/sd -name "*.db" | head -1); python $R/tools/launch_sequence.py $DB > $O/demo_sequence.txt 2>&1; head -1 $O/demo_sequence.txt
rm -rf $O/sd
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1); python $R/tools/launch_sequence.py $DB2 > $O/headline_sequence.txt 2>&1; head -1 $O/headline_sequence.txt
rm -rf $O/ss
