# round 5, GPU call 1: new tests, FPS geometry sweep, stress (both arithmetic modes), 128^3 per-round study, demo workload
mkdir -p gpurun_out/r5c1
O=$PWD/gpurun_out/r5c1
R=$PWD
timeout 480 python -m pytest tests/test_gpu_fps_abort.py tests/test_gpu_ops.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 150 python tools/fps_sweep.py > $O/fps_sweep.txt 2>&1; cat $O/fps_sweep.txt
timeout 240 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/bench_stress.err; tail -c 600 $O/bench_stress.err
show() { python - <<P
import json
try:
    d=json.loads(open("$O/$1").read().strip().splitlines()[-1])
    print("$1 value %.4g ms/step %.2f frac %.4f failed %d single %s"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"], (d.get("single_scene") or {}).get("ms_per_scene")), " ".join("r%d %.2f/%.0fTF"%(r["round"],r["avg_launch_ms"],r["achieved"]) for r in (d["roofline"].get("per_round") or [])))
    for k in ("stage_ms_per_scene",): print(k, d.get(k))
    print("cfg", {k:v for k,v in d["config"].items() if k in ("proposals_per_scene","queries_per_scene","objectness_bias_shift","proposals_kept_scene0","graph_detect","scenes_done")})
    print("single", d.get("single_scene"))
except Exception as e: print("$1 ERR", e, open("$O/$2").read()[-900:])
P
}
timeout 300 python bench.py --config mise128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras > $O/m128_if4.json 2> $O/m128_if4.err; show m128_if4.json m128_if4.err
timeout 300 python bench.py --config mise128 --steps 6 --warmup 2 --in-flight 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if1.json 2> $O/m128_if1.err; show m128_if1.json m128_if1.err
timeout 200 python bench.py --config demo --steps 8 --warmup 3 --no-cpu-baseline --no-extras > $O/demo.json 2> $O/demo.err; show demo.json demo.err
timeout 200 python bench.py --config demo --steps 8 --warmup 3 --no-cpu-baseline --no-extras --graph-detect > $O/demo_graph.json 2> $O/demo_graph.err; show demo_graph.json demo_graph.err; tail -3 $O/demo_graph.err
timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > $O/headline.json 2> $O/headline.err; show headline.json headline.err
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $O/tr -o tr -- python $R/bench.py --config mise128 --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-latency > $O/m128_trace.json 2> $O/m128_trace.err
DB=$(find $O/tr -name "*.db" | head -1); echo DB $DB
python $R/tools/decoder_rounds.py $DB > $O/decoder_rounds_if4.txt 2>&1; head -50 $O/decoder_rounds_if4.txt
python $R/tools/gpu_timeline.py $DB > $O/gpu_timeline_m128.txt 2>&1; head -45 $O/gpu_timeline_m128.txt
python $R/tools/rocpd_stats.py $DB 40 > $O/m128_kernel_trace.txt 2>&1
rm -rf $O/tr
