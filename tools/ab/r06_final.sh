# round 6, measurement artefacts on the final tree: one gpurun call.
#   bash tools/ab/r06_final.sh            -> gpurun_out/r6final/*  (copied into profiles/ by hand, see tools/ab/r06_sessions.md)
R=$PWD; O=$PWD/gpurun_out/r6final; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 60 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py --steps 20 --warmup 5 > $O/bench_headline.json 2> $O/bench_headline.err
timeout 300 python bench.py --config stress --steps 3 --warmup 1 > $O/bench_stress.json 2> $O/bench_stress.err
timeout 600 python bench.py --config mise128 --steps 12 --warmup 2 > $O/bench_mise128.json 2> $O/bench_mise128.err
timeout 300 python bench.py --config dense32 --steps 6 --warmup 2 > $O/bench_dense32.json 2> $O/bench_dense32.err
timeout 300 python bench.py --config demo --steps 40 --warmup 5 > $O/bench_demo.json 2> $O/bench_demo.err
python - <<P
import json
for f in ("headline","stress","mise128","dense32","demo"):
    try:
        d=json.loads(open("$O/bench_%s.json"%f).read().strip().splitlines()[-1])
        cb=d.get("cpu_baseline") or {}
        print(f, "value %.4g %s ms/step %.2f frac %.4f failed %d single %s hbm %s cpu_baseline %s (%s cores) iou %s" % (
            d["value"], d["unit"], d["ms_per_step"], d["roofline"]["frac"], d["config"]["scenes_failed"],
            (d.get("single_scene") or {}).get("ms_per_scene"), d["config"].get("hbm_peak_gib"), cb.get("value"), cb.get("cores"),
            d["config"].get("parity_iou")))
    except Exception as e: print(f, "ERR", e)
P
cd /tmp; export TMPDIR=/tmp
# steady-state headline under the tracer: per-kernel table + GPU timeline
timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 > $O/bench_profiled.json 2> $O/bench_profiled.err
DB=$(find $O/kt -name "*.db" | head -1)
python $R/tools/rocpd_stats.py $DB 45 > $O/bench_kernel_trace.txt 2>&1; head -6 $O/bench_kernel_trace.txt | cut -c1-150
python $R/tools/gpu_timeline.py $DB > $O/gpu_timeline.txt 2>&1; head -6 $O/gpu_timeline.txt
rm -rf $O/kt
# one scene at a time: per-kernel table + ordered launch list
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/bench_ss.json 2> $O/bench_ss.err
DB2=$(find $O/ss -name "*.db" | head -1)
python $R/tools/launch_sequence.py $DB2 > $O/single_scene_sequence.txt 2>&1; head -2 $O/single_scene_sequence.txt
python - <<P
# per-kernel table of the last COMPLETE scene (between the last two SA1 FPS launches)
import sqlite3, re
con = sqlite3.connect("$DB2")
rows = con.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
st = [i for i, r in enumerate(rows) if "fps_kernelILi10" in r[0]]
seq = rows[st[-2]:st[-1]]
agg = {}
for n, s, e in seq:
    a = agg.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
tot = sum(v[1] for v in agg.values())
with open("$O/single_scene_kernel_trace.txt", "w") as f:
    f.write("# one scene at a time, the last complete scene: %d launches, %.3f ms busy, %.3f ms first-to-last\n" % (len(seq), tot / 1e6, (max(r[2] for r in seq) - seq[0][1]) / 1e6))
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
        f.write("%-100s %5d %10.3f ms %6.2f %%\n" % (n[:100], v[0], v[1] / 1e6, 100.0 * v[1] / tot))
print(open("$O/single_scene_kernel_trace.txt").read()[:1500])
P
rm -rf $O/ss
# HBM traffic of the decoder launches inside the benchmark: FETCH_SIZE and WRITE_SIZE in separate passes
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
d=${c%%:*}; ctr=${c##*:}
timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/traffic/$d -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 2 --warmup 1 > $O/bench_pmc_$d.json 2> $O/bench_pmc_$d.err
done
# matrix-pipe busy and friends, decoder alone (F16X3 and F16X1)
n=0
for cs in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
n=$((n+1))
timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/sq3/$n -- python $R/tools/dec_only.py 3 > /dev/null 2> $O/sq3_$n.err
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq1/1 -- python $R/tools/dec_only.py 1 > /dev/null 2> $O/sq1_1.err
# the frag-rows encoder GEMM: the same counters
n=0
for cs in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "FETCH_SIZE" "WRITE_SIZE"; do
n=$((n+1))
timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/sqg/$n -- python $R/tools/gemm_frag_bench.py > /dev/null 2> $O/sqg_$n.err
done
cd $R
python - <<P > $O/npoints.txt
import json
d = json.loads(open("$O/bench_pmc_fetch.json").read().strip().splitlines()[-1])
print(int(3 * d["config"]["queries_per_scene"]))        # the three scenes' nine launches (pmc_traffic.py --first 9)
P
python tools/pmc_traffic.py --first 9 $O/traffic $(cat $O/npoints.txt) r06 > $O/decoder_traffic.txt 2>&1; cat $O/decoder_traffic.txt
python tools/pmc_sq.py $O/sq3 "occ_decode8_kernelILi3E" --json f16x3 "profiles/r06_decoder8_pmc.txt (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... over tools/dec_only.py 3 / 1, round 6 final call; busy = MFMA_BUSY / (4 x WAVE_CYCLES / 2), tools/pmc_sq.py)" > $O/decoder8_pmc_f16x3.txt 2>&1; cat $O/decoder8_pmc_f16x3.txt
python tools/pmc_sq.py $O/sq1 "occ_decode8_kernelILi1E" --json f16x1 "profiles/r06_decoder8_pmc.txt (rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES ... over tools/dec_only.py 3 / 1, round 6 final call; busy = MFMA_BUSY / (4 x WAVE_CYCLES / 2), tools/pmc_sq.py)" > $O/decoder8_pmc_f16x1.txt 2>&1; cat $O/decoder8_pmc_f16x1.txt
python tools/pmc_sq.py $O/sqg "gemm_rowsf_kernel" > $O/gemm_rowsf_pmc.txt 2>&1; cat $O/gemm_rowsf_pmc.txt
python tools/pmc_sq.py $O/sqg "gemm_rows8_kernel" > $O/gemm_rows8_pmc.txt 2>&1; cat $O/gemm_rows8_pmc.txt
cp profiles/decoder_traffic.json profiles/decoder_mfma_busy.json $O/ 2>/dev/null
rm -rf $O/traffic $O/sq3 $O/sq1 $O/sqg
if [ -d .r05tree ]; then bash tools/ab/r06_ab_r05.sh > $O/ab_r05.txt 2>&1; cat $O/ab_r05.txt | cut -c1-60; fi
du -sh $O
