# round 5, final record (same as call 10, on the tree with the faster FPS): full GPU suite, every bench configuration, sweep, profiled trace
mkdir -p gpurun_out/r5c15
O=$PWD/gpurun_out/r5c15
R=$PWD
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --stats-out $O/scene_stats.json > $O/bench_headline.json 2> $O/bench_headline.err; tail -c 300 $O/bench_headline.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_headline_20steps.json 2> $O/err.txt
timeout 300 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/err.txt
timeout 400 python bench.py --config mise128 --steps 4 --warmup 1 > $O/bench_mise128.json 2> $O/err.txt
timeout 300 python bench.py --config dense32 --steps 4 --warmup 1 > $O/bench_dense32.json 2> $O/err.txt
timeout 300 python bench.py --config demo --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_demo.json 2> $O/err.txt
timeout 400 python bench.py --config mise128 --scenes 311 --warmup 1 --no-cpu-baseline --no-extras --no-latency > $O/bench_sweep311_1gpu.json 2> $O/sweep.err; tail -c 300 $O/sweep.err
python - <<P
import json
for f in ("headline","headline_20steps","stress","mise128","dense32","demo","sweep311_1gpu"):
    try:
        d=json.loads(open("$O/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value %.4g %s ms/step %.2f frac %.4f failed %d done %d single %s"%(d["value"],d["unit"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"],d["config"]["scenes_done"],(d.get("single_scene") or {}).get("ms_per_scene")), "fps", (d.get("roofline_fps") or {}).get("avg_launch_ms"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "iou", d["config"].get("parity_iou"))
    except Exception as e: print(f, "ERR", e)
P
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -8 $O/bench_kernel_trace.txt | cut -c1-170
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB2 --last-scene > $O/single_scene_kernel_trace.txt 2>&1; head -30 $O/single_scene_kernel_trace.txt | cut -c1-170
rm -rf $O/kt $O/ss
