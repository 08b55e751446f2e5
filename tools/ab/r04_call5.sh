mkdir -p gpurun_out/r4c5
O=gpurun_out/r4c5
timeout 300 python -m pytest tests/test_gpu_generator.py -m gpu -q -p no:cacheprovider -k "bands" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 3"
run() { name=$1; shift; timeout 300 $B "$@" > $O/bench_$name.json 2> $O/bench_$name.err; python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.3f ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]), " ".join("r%d %.2f"%(r["round"],r["avg_launch_ms"]) for r in d["roofline"]["per_round"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-700:])
P
}
run split0_a
run split8_a --cu-split 8
run split16_a --cu-split 16
run split32_a --cu-split 32
run split0_b
run split8_b --cu-split 8
run split16_b --cu-split 16
run split32_b --cu-split 32
run split24 --cu-split 24
run split0_c
