mkdir -p gpurun_out/r4c9
O=$PWD/gpurun_out/r4c9
R=$PWD
run() { name=$1; dir=$2; shift; shift; (cd $dir && timeout 200 python bench.py --no-cpu-baseline --no-latency "$@" > $O/bench_$name.json 2> $O/bench_$name.err); python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f avg_launch %.3f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["avg_launch_ms"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
for i in 1 2 3; do
run r03_head_$i $R/.r03tree --steps 8 --warmup 3
run r04_head_$i $R --steps 8 --warmup 3
done
run r03_m128 $R/.r03tree --config mise128 --steps 4 --warmup 2
run r04_m128 $R --config mise128 --steps 4 --warmup 2
run r03_dense32 $R/.r03tree --config dense32 --steps 4 --warmup 2
run r04_dense32 $R --config dense32 --steps 4 --warmup 2
run r03_stress $R/.r03tree --config stress --steps 3 --warmup 1
run r04_stress $R --config stress --steps 3 --warmup 1
