mkdir -p gpurun_out/r4c7
O=gpurun_out/r4c7
timeout 300 python -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider -k "claimed" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 3"
run() { name=$1; shift; env "$@" timeout 200 $B $EXTRA > $O/bench_$name.json 2> $O/bench_$name.err; python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]), " ".join("r%d %.2f"%(r["round"],r["avg_launch_ms"]) for r in d["roofline"]["per_round"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-700:])
P
}
EXTRA=""
run persist_a RFD_DECODER_CHUNK=0
run chunk16_a RFD_DECODER_CHUNK=16
run chunk8_a RFD_DECODER_CHUNK=8
run chunk32_a RFD_DECODER_CHUNK=32
run chunk64_a RFD_DECODER_CHUNK=64
run persist_b RFD_DECODER_CHUNK=0
run chunk16_b RFD_DECODER_CHUNK=16
run chunk32_b RFD_DECODER_CHUNK=32
EXTRA="--config stress --steps 3 --warmup 1"
run stress_persist RFD_DECODER_CHUNK=0
run stress_chunk16 RFD_DECODER_CHUNK=16
run stress_chunk32 RFD_DECODER_CHUNK=32
EXTRA="--config mise128 --steps 4 --warmup 2"
run m128_persist RFD_DECODER_CHUNK=0
run m128_chunk16 RFD_DECODER_CHUNK=16
