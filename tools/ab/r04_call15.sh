mkdir -p gpurun_out/r4c15
O=$PWD/gpurun_out/r4c15
run() { name=$1; shift; (env "$@" timeout 200 python bench.py --no-cpu-baseline --no-latency --no-extras --steps 8 --warmup 3 $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err); python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
EXTRA=""
for i in 1 2; do
run q_default_$i A=1
run q2_$i GPU_MAX_HW_QUEUES=2
run q8_$i GPU_MAX_HW_QUEUES=8
run q16_$i GPU_MAX_HW_QUEUES=16
done
EXTRA="--in-flight 4"
run q16_if4 GPU_MAX_HW_QUEUES=16
run qdef_if4 A=1
EXTRA="--in-flight 2"
run q16_if2 GPU_MAX_HW_QUEUES=16
run qdef_if2 A=1
