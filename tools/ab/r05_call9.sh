# round 5, GPU call 9: scenes in flight, 128^3 and headline (same box)
mkdir -p gpurun_out/r5c9
O=$PWD/gpurun_out/r5c9
show() { python - <<P
import json
try:
    d=json.loads(open("$O/$1").read().strip().splitlines()[-1])
    print("$1 value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]), " ".join("r%d %.2f/%.0fTF"%(r["round"],r["avg_launch_ms"],r["achieved"]) for r in (d["roofline"].get("per_round") or [])))
except Exception as e: print("$1 ERR", e, open("$O/$2").read()[-900:])
P
}
for fl in 4 6 8 4 6 8 12; do
timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --in-flight $fl --no-latency --no-cpu-baseline --no-extras > $O/m128_fl${fl}.json 2> $O/err.txt; show m128_fl${fl}.json err.txt
done
for fl in 4 5 6 8 4 6; do
timeout 300 python bench.py --steps 8 --warmup 3 --in-flight $fl --no-latency --no-cpu-baseline --no-extras > $O/head_fl${fl}.json 2> $O/err.txt; show head_fl${fl}.json err.txt
done
