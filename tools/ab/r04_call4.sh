mkdir -p gpurun_out/r4c4
O=gpurun_out/r4c4
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 3"
run() { name=$1; shift; env "$@" timeout 300 $B > $O/bench_$name.json 2> $O/bench_$name.err; python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.3f ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
run shared_early A=1
run replicas_early RFD_BENCH_REPLICAS=1
run shared_late RFD_STATUS_LATE=1
run replicas_late RFD_BENCH_REPLICAS=1 RFD_STATUS_LATE=1
run shared_early2 A=1
run replicas_late2 RFD_BENCH_REPLICAS=1 RFD_STATUS_LATE=1
