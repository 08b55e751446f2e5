mkdir -p gpurun_out/r4c2
O=gpurun_out/r4c2
RFD_BANDS_OUT=$O/logit_bands.txt timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_generator.py -m gpu -q -s -p no:cacheprovider > $O/pytest.txt 2>&1
tail -3 $O/pytest.txt; grep "codes x" $O/pytest.txt
cat $O/logit_bands.txt
timeout 600 python tools/ab/energy_ledger.py $O/energy.txt > $O/energy.log 2>&1
cat $O/energy.txt
B="python bench.py --no-cpu-baseline --no-latency --steps 8 --warmup 3"
for cu in 256 248 240 232 224 256; do
RFD_DECODER_CUS=$cu timeout 300 $B > $O/bench_cu$cu.json 2> $O/bench_cu$cu.err
python - <<P
import json
d=json.loads(open("$O/bench_cu$cu.json").read().strip().splitlines()[-1])
print("CUS $cu value %.3f ms/step %.2f frac %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), " ".join("r%d %.2fms"%(r["round"],r["avg_launch_ms"]) for r in d["roofline"]["per_round"]))
P
done
