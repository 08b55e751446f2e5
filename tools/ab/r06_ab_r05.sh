# round-5 HEAD (337100b, a git worktree under .r05tree/) against this tree on ONE box, alternating: the boxes differ by
# ~3 % in decoder speed, so only a same-box comparison says what round 6 changed.
# Set-up (here, before the gpurun call): git worktree add .r05tree 337100b && (cd .r05tree && python -m rfdnet_amd.build);
# afterwards: git worktree remove .r05tree --force   (the directory is git-ignored but travels with the snapshot)
O=$PWD/gpurun_out/r6ab; mkdir -p $O; R=$PWD
for i in 1 2 3; do
(cd $R/.r05tree && timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 4 > $O/r05_$i.json 2>/dev/null)
(cd $R && timeout 200 python bench.py --no-cpu-baseline --no-extras --steps 16 --warmup 4 > $O/r06_$i.json 2>/dev/null)
done
python - <<P
import json
for n in ("r05", "r06"):
    for i in (1, 2, 3):
        try:
            d = json.loads(open("$O/%s_%d.json" % (n, i)).read().strip().splitlines()[-1])
            print(n, i, "value %.3f frac %.4f single %.2f ms" % (d["value"], d["roofline"]["frac"], d["single_scene"]["ms_per_scene"]), d["single_scene"]["stage_ms"])
        except Exception as e:
            print(n, i, "ERR", e)
P
