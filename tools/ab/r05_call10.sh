# round 5, GPU call 10: the record of the final tree -- full GPU suite, every bench configuration, the 311-scene sweep on one
# GPU, profiled kernel trace, preflight on hardware
mkdir -p gpurun_out/r5c10
O=$PWD/gpurun_out/r5c10
R=$PWD
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 400 python bench.py --stats-out $O/scene_stats.json > $O/bench_headline.json 2> $O/bench_headline.err; tail -c 300 $O/bench_headline.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_headline_20steps.json 2> $O/err.txt
timeout 300 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/err.txt
timeout 400 python bench.py --config mise128 --steps 4 --warmup 1 > $O/bench_mise128.json 2> $O/err.txt
timeout 300 python bench.py --config dense32 --steps 4 --warmup 1 > $O/bench_dense32.json 2> $O/err.txt
timeout 300 python bench.py --config demo --steps 8 --warmup 3 --no-cpu-baseline > $O/bench_demo.json 2> $O/err.txt
timeout 400 python bench.py --config mise128 --scenes 311 --warmup 1 --no-cpu-baseline --no-extras --no-latency --stats-out $O/sweep311_stats.json > $O/bench_sweep311_1gpu.json 2> $O/sweep.err; tail -c 300 $O/sweep.err
python - <<P
import json
for f in ("headline","headline_20steps","stress","mise128","dense32","demo","sweep311_1gpu"):
    try:
        d=json.loads(open("$O/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value %.4g %s ms/step %.2f frac %.4f failed %d done %d single %s"%(d["value"],d["unit"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"],d["config"]["scenes_done"],(d.get("single_scene") or {}).get("ms_per_scene")), "busy", d["roofline"].get("mfma_busy"), "cpu", (d.get("cpu_baseline") or {}).get("value"), "iou", d["config"].get("parity_iou"))
    except Exception as e: print(f, "ERR", e)
P
timeout 60 python bench.py --preflight > $O/preflight_1.json 2> $O/preflight_1.err; cat $O/preflight_1.json $O/preflight_1.err | tail -3
RFD_BENCH_ONE_DEVICE=1 timeout 100 python bench.py --preflight --gpus 2 > $O/preflight_2_onedev.json 2> $O/preflight_2.err; cat $O/preflight_2_onedev.json | tail -1; tail -2 $O/preflight_2.err
RFD_BENCH_FORCE_DIST=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --preflight --gpus 1 > $O/preflight_rccl1.json 2> $O/preflight_rccl1.err; tail -1 $O/preflight_rccl1.json; tail -2 $O/preflight_rccl1.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -8 $O/bench_kernel_trace.txt | cut -c1-170
rm -rf $O/kt
