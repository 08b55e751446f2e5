# round 5, GPU call 8: does a smaller persistent decoder grid pay at 128^3, where 45 ms of a scene are NOT the decoder?
mkdir -p gpurun_out/r5c8
O=$PWD/gpurun_out/r5c8
show() { python - <<P
import json
try:
    d=json.loads(open("$O/$1").read().strip().splitlines()[-1])
    print("$1 value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]), " ".join("r%d %.2f/%.0fTF"%(r["round"],r["avg_launch_ms"],r["achieved"]) for r in (d["roofline"].get("per_round") or [])))
except Exception as e: print("$1 ERR", e, open("$O/$2").read()[-900:])
P
}
for cus in 0 240 224 208 192 0 232 216; do
RFD_DECODER_CUS=$cus timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_cus$cus.json 2> $O/err.txt; show m128_cus$cus.json err.txt
done
for fl in 2 3 6; do
timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --in-flight $fl --no-latency --no-cpu-baseline --no-extras > $O/m128_fl$fl.json 2> $O/err.txt; show m128_fl$fl.json err.txt
done
