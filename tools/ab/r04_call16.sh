mkdir -p gpurun_out/r4c16
O=$PWD/gpurun_out/r4c16
run() { name=$1; shift; (env "$@" timeout 200 python bench.py --no-cpu-baseline --no-latency --no-extras --steps 6 --warmup 3 $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err); python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
for q in 8 16 32; do for f in 3 4 5 6; do EXTRA="--in-flight $f"; run q${q}_if$f GPU_MAX_HW_QUEUES=$q; done; done
EXTRA="--in-flight 4"; run q16_if4_b GPU_MAX_HW_QUEUES=16
EXTRA="--in-flight 5"; run q16_if5_b GPU_MAX_HW_QUEUES=16
EXTRA="--in-flight 3"; run qdef_if3_b A=1
