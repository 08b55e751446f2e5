# round 5, GPU call 3: FPS abort tests with the static-LDS CU holder; MISE parity after the subdivide skips; 128^3 before/after
mkdir -p gpurun_out/r5c3
O=$PWD/gpurun_out/r5c3
R=$PWD
timeout 600 python -m pytest tests/test_gpu_fps_abort.py tests/test_gpu_generator.py tests/test_gpu_fullsize.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
show() { python - <<P
import json
try:
    d=json.loads(open("$O/$1").read().strip().splitlines()[-1])
    print("$1 value %.4g ms/step %.2f frac %.4f failed %d single %s"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["config"]["scenes_failed"], (d.get("single_scene") or {}).get("ms_per_scene")), " ".join("r%d %.2f/%.0fTF"%(r["round"],r["avg_launch_ms"],r["achieved"]) for r in (d["roofline"].get("per_round") or [])))
    print("  single", d.get("single_scene"), "parity", d["config"].get("parity_iou"))
except Exception as e: print("$1 ERR", e, open("$O/$2").read()[-900:])
P
}
timeout 300 python bench.py --config mise128 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/m128_if4.json 2> $O/m128_if4.err; show m128_if4.json m128_if4.err
timeout 300 python bench.py --config mise128 --steps 6 --warmup 2 --in-flight 1 --no-latency --no-cpu-baseline --no-extras > $O/m128_if1.json 2> $O/m128_if1.err; show m128_if1.json m128_if1.err
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/headline.json 2> $O/headline.err; show headline.json headline.err
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --config mise128 --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/m128_ss.json 2> $O/m128_ss.err
DB=$(find $O/ss -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB --last-scene > $O/m128_single_scene_kernel_trace.txt 2>&1; head -24 $O/m128_single_scene_kernel_trace.txt | cut -c1-175
python - <<P
import sqlite3
con=sqlite3.connect("$DB")
rows=con.execute("select s.kernel_name,d.start,d.end,d.grid_size_x,d.grid_size_y from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like '%subdivide%' or s.kernel_name like '%collect%' order by d.start").fetchall()
for n,s,e,gx,gy in rows[-16:]: print("%-40s %9.3f ms grid %d x %d"%(n[17:57],(e-s)/1e6,gx,gy))
P
rm -rf $O/ss
