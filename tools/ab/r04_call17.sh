mkdir -p gpurun_out/r4c17
O=$PWD/gpurun_out/r4c17
run() { name=$1; shift; (env GPU_MAX_HW_QUEUES=16 "$@" timeout 200 python bench.py --no-cpu-baseline --no-latency --no-extras --steps 6 --warmup 3 --in-flight 4 $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err); python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
EXTRA=""
for i in 1 2; do
run base_$i A=1
run cus248_$i RFD_DECODER_CUS=248
run cus240_$i RFD_DECODER_CUS=240
run chunk32_$i RFD_DECODER_CHUNK=32
run static_$i RFD_DECODER_STATIC=1
done
EXTRA="--blit-round 1"; run blit1 A=1
EXTRA="--blit-round -1"; run blitnow A=1
EXTRA="--blit-round 0"; run blit0 A=1
