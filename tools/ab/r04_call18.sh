mkdir -p gpurun_out/r4c18
O=$PWD/gpurun_out/r4c18
R=$PWD
timeout 600 python -m pytest tests/test_gpu_bench_dist.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
timeout 400 python bench.py --stats-out $O/scene_stats.json > $O/bench_headline.json 2> $O/bench_headline.err; tail -2 $O/bench_headline.err
timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_headline_8steps.json 2> $O/bench_headline_8steps.err
timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_headline_20steps.json 2> $O/bench_headline_20steps.err
timeout 500 python bench.py --config mise128 --steps 3 --warmup 1 > $O/bench_mise128.json 2> $O/bench_mise128.err; tail -2 $O/bench_mise128.err
timeout 300 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/bench_stress.err
timeout 300 python bench.py --config dense32 --steps 3 --warmup 1 > $O/bench_dense32.json 2> $O/bench_dense32.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -8 $O/bench_kernel_trace.txt
timeout 400 rocprofv3 --kernel-trace --memory-copy-trace -d $O/tl -o tl -- python $R/bench.py --no-cpu-baseline --no-latency --no-extras --steps 6 --warmup 2 > $O/bench_tl.json 2> $O/bench_tl.err
DB=$(find $O/tl -name "*.db" | head -1); python $R/tools/gpu_timeline.py $DB > $O/gpu_timeline.txt 2>&1; head -40 $O/gpu_timeline.txt
cd $R
for i in 1 2 3; do
(cd $R/.r03tree && timeout 200 python bench.py --no-cpu-baseline --no-latency --steps 6 --warmup 3 > $O/ab_r03_$i.json 2>/dev/null)
(cd $R && timeout 200 python bench.py --no-cpu-baseline --no-latency --no-extras --steps 6 --warmup 3 > $O/ab_r04_$i.json 2>/dev/null)
done
python - <<P
import json
for f in ("headline","headline_8steps","headline_20steps","mise128","stress","dense32","profiled"):
    try:
        d=json.loads(open("$O/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f,"value %.4g ms/step %.2f frac %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), "failed", d["config"].get("scenes_failed"), "in flight", d["config"].get("scenes_in_flight_per_gpu"))
    except Exception as e: print(f,"ERR",e)
for n in ("r03","r04"):
    for i in (1,2,3):
        try:
            d=json.loads(open("$O/ab_%s_%d.json"%(n,i)).read().strip().splitlines()[-1]); print(n,i,"value %.3f frac %.4f"%(d["value"],d["roofline"]["frac"]))
        except Exception as e: print(n,i,"ERR",e)
P
rm -rf $O/kt $O/tl
