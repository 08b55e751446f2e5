mkdir -p gpurun_out/r4c1
O=gpurun_out/r4c1
RFD_BANDS_OUT=$O/logit_bands.txt timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_generator.py -m gpu -q -s -p no:cacheprovider -x > $O/pytest.txt 2>&1
tail -5 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 3"
timeout 300 $B > $O/bench_claim.json 2> $O/bench_claim.err
RFD_DECODER_STATIC=1 timeout 300 $B > $O/bench_static.json 2> $O/bench_static.err
timeout 300 $B --blit-round 1 > $O/bench_claim_blit1.json 2> $O/bench_claim_blit1.err
timeout 300 $B --blit-round -1 > $O/bench_claim_blitnow.json 2> $O/bench_claim_blitnow.err
timeout 300 $B > $O/bench_claim2.json 2> $O/bench_claim2.err
RFD_DECODER_STATIC=1 timeout 300 $B > $O/bench_static2.json 2> $O/bench_static2.err
timeout 300 $B --config mise128 --steps 4 --warmup 2 > $O/bench_m128_claim.json 2> $O/bench_m128_claim.err
RFD_DECODER_STATIC=1 timeout 300 $B --config mise128 --steps 4 --warmup 2 > $O/bench_m128_static.json 2> $O/bench_m128_static.err
timeout 300 $B --config stress --steps 4 --warmup 1 > $O/bench_stress_claim.json 2> $O/bench_stress_claim.err
RFD_DECODER_STATIC=1 timeout 300 $B --config stress --steps 4 --warmup 1 > $O/bench_stress_static.json 2> $O/bench_stress_static.err
for f in claim static claim_blit1 claim_blitnow claim2 static2 m128_claim m128_static stress_claim stress_static; do echo $f; python - <<P
import json
try:
    d=json.loads(open("$O/bench_$f.json").read().strip().splitlines()[-1])
    print(" value %.3f ms/step %.2f frac %.4f avg_launch %.3f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["avg_launch_ms"]))
    for r in d["roofline"]["per_round"] or []: print("   round %d launches %d real %d ms/launch %.3f TF %.1f"%(r["round"],r["launches"],r["real_points"],r["avg_launch_ms"],r["achieved"]))
except Exception as e: print(" ERR",e); print(open("$O/bench_$f.err").read()[-800:])
P
done
cat $O/logit_bands.txt
