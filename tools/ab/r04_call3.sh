mkdir -p gpurun_out/r4c3
O=gpurun_out/r4c3
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1
tail -15 $O/pytest.txt
timeout 600 python bench.py --steps 8 --warmup 3 > $O/bench_headline.json 2> $O/bench_headline.err
tail -3 $O/bench_headline.err
python - <<P
import json
d=json.loads(open("$O/bench_headline.json").read().strip().splitlines()[-1])
print("value %.3f ms/step %.2f frac %.4f"%(d["value"],d["ms_per_step"],d["roofline"]["frac"]))
print(json.dumps(d.get("roofline_fps"))[:600]); print(json.dumps(d.get("roofline_ball_query"))[:900])
print(d["cpu_baseline"]["stage_s"], d["cpu_baseline"]["sample"][-400:])
print(d.get("single_scene"), d["config"]["decoder_selfcheck"], d.get("parity",{}).get("min_iou"))
P
timeout 600 python tools/kbench.py > $O/kbench.json 2> $O/kbench.err; tail -3 $O/kbench.err; head -c 1500 $O/kbench.json
