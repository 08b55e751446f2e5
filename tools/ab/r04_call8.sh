mkdir -p gpurun_out/r4c8
O=$PWD/gpurun_out/r4c8
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
timeout 400 python bench.py --stats-out $O/scene_stats.json > $O/bench_headline.json 2> $O/bench_headline.err; tail -2 $O/bench_headline.err
timeout 200 python bench.py --no-cpu-baseline --steps 8 --warmup 3 > $O/bench_headline_8steps.json 2> $O/bench_headline_8steps.err
timeout 200 python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $O/bench_headline_20steps.json 2> $O/bench_headline_20steps.err
timeout 500 python bench.py --config mise128 --steps 3 --warmup 1 > $O/bench_mise128.json 2> $O/bench_mise128.err; tail -2 $O/bench_mise128.err
timeout 300 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/bench_stress.err; tail -2 $O/bench_stress.err
timeout 300 python bench.py --config dense32 --steps 3 --warmup 1 > $O/bench_dense32.json 2> $O/bench_dense32.err; tail -2 $O/bench_dense32.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 3 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -12 $O/bench_kernel_trace.txt
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB2 --last-scene > $O/single_scene_kernel_trace.txt 2>&1; head -30 $O/single_scene_kernel_trace.txt
for c in FETCH_SIZE WRITE_SIZE; do
timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/traffic/$c -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 2 --warmup 1 > $O/bench_pmc_$c.json 2> $O/bench_pmc_$c.err
done
for cs in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
n=$(echo $cs | cut -c1-20 | tr ' ' '_')
timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/sq/$n -- python $R/tools/dec_only.py 3 > /dev/null 2> $O/sq_$n.err
done
python $R/tools/pmc_sq.py $O/sq > $O/decoder8_pmc.txt 2>&1; cat $O/decoder8_pmc.txt
cd $R
find $O/traffic -name "*counter_collection.csv" | head; find $O/traffic -name "*kernel_trace.csv" | head -3
python - <<P
import json,glob
for f in ("headline","headline_8steps","headline_20steps","mise128","stress","dense32","profiled"):
    try:
        d=json.loads(open("$O/bench_%s.json"%f).read().strip().splitlines()[-1])
        print(f,"value %.4g ms/step %.2f frac %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]), "failed", d["config"].get("scenes_failed"))
    except Exception as e: print(f,"ERR",e)
P
rm -rf $O/kt $O/ss
du -sh $O
