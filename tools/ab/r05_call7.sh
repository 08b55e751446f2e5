# round 5, GPU call 7: same-box A/B of the MISE subdivision skips (round-4 mise.hip as a side library), 128^3 and headline
mkdir -p gpurun_out/r5c7
O=$PWD/gpurun_out/r5c7
show() { python - <<P
import json
try:
    d=json.loads(open("$O/$1").read().strip().splitlines()[-1])
    print("$1 value %.4g ms/step %.2f frac %.4f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"]), " ".join("r%d %.2f/%.0fTF"%(r["round"],r["avg_launch_ms"],r["achieved"]) for r in (d["roofline"].get("per_round") or [])))
except Exception as e: print("$1 ERR", e, open("$O/$2").read()[-900:])
P
}
V=$PWD/rfdnet_amd/lib/variants/librfd_miser04.so
for rep in a b; do
RFD_HIP_LIB=$V timeout 300 python bench.py --config mise128 --steps 6 --warmup 2 --in-flight 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if1_r04_$rep.json 2> $O/err.txt; show m128_if1_r04_$rep.json err.txt
timeout 300 python bench.py --config mise128 --steps 6 --warmup 2 --in-flight 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if1_r05_$rep.json 2> $O/err.txt; show m128_if1_r05_$rep.json err.txt
RFD_HIP_LIB=$V timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if4_r04_$rep.json 2> $O/err.txt; show m128_if4_r04_$rep.json err.txt
timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if4_r05_$rep.json 2> $O/err.txt; show m128_if4_r05_$rep.json err.txt
done
RFD_HIP_LIB=$V timeout 300 python bench.py --steps 10 --warmup 3 --no-latency --no-cpu-baseline --no-extras > $O/head_r04.json 2> $O/err.txt; show head_r04.json err.txt
timeout 300 python bench.py --steps 10 --warmup 3 --no-latency --no-cpu-baseline --no-extras > $O/head_r05.json 2> $O/err.txt; show head_r05.json err.txt
