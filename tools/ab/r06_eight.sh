# eight-GPU readiness without eight GPUs (one MI355X box).
#  * the N = 8 code path with every rank on device 0 (gloo for the one collective): launcher, sharding, NUMA logic,
#    all-gather of the statistics, JSON line
#  * --preflight --gpus 8: must FAIL fast on a 1-GPU box with one actionable line (not hang), and pass in one-device mode
#  * host CPU seconds per scene (headline + demo): how many cores 8 ranks x 4 scene threads need
O=$PWD/gpurun_out/r6eight; mkdir -p $O
RFD_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 8 --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-latency --no-extras > $O/eight_dryrun.json 2> $O/eight_dryrun.err; echo "dry run rc=$?"; tail -c 400 $O/eight_dryrun.json; echo
timeout 120 python bench.py --preflight --gpus 8 > $O/preflight_8.json 2> $O/preflight_8.err; echo "preflight 8 on one GPU rc=$? (expected non-zero)"; tail -2 $O/preflight_8.err
RFD_BENCH_ONE_DEVICE=1 timeout 120 python bench.py --preflight --gpus 8 > $O/preflight_8_onedev.json 2> $O/preflight_8_onedev.err; echo "preflight 8 one-device rc=$?"; tail -c 600 $O/preflight_8_onedev.json; echo
timeout 120 python bench.py --preflight > $O/preflight_1.json 2> $O/preflight_1.err; echo "preflight 1 rc=$?"; tail -c 400 $O/preflight_1.json; echo
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-latency > $O/headline_cpu.json 2> $O/headline_cpu.err
timeout 300 python bench.py --config demo --steps 40 --warmup 5 --no-cpu-baseline --no-extras --no-latency > $O/demo_cpu.json 2> $O/demo_cpu.err
python - <<P
import json, os
for f in ("headline_cpu", "demo_cpu"):
    d = json.loads(open("$O/%s.json" % f).read().strip().splitlines()[-1])
    c = d["config"]
    print(f, "value %.2f %s, host CPU s per scene %.4f, => cores busy at this rate %.2f, hbm peak %.1f GiB, cores in mask %d" % (
        d["value"], d["unit"], c["host_cpu_s_per_scene_rank0"], c["host_cpu_s_per_scene_rank0"] * d["value"], c["hbm_peak_gib"], len(os.sched_getaffinity(0))))
P
