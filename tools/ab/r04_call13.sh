mkdir -p gpurun_out/r4c13
O=$PWD/gpurun_out/r4c13
R=$PWD
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py tests/test_gpu_net.py tests/test_gpu_sa_fused.py tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
run() { name=$1; dir=$2; shift; shift; (cd $dir && env $ENVX timeout 200 python bench.py --no-cpu-baseline --no-latency "$@" > $O/bench_$name.json 2> $O/bench_$name.err); python - <<P
import json
try:
    d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1])
    print("$name value %.4g ms/step %.2f frac %.4f avg_launch %.3f failed %d"%(d["value"],d["ms_per_step"],d["roofline"]["frac"],d["roofline"]["avg_launch_ms"],d["config"]["scenes_failed"]))
except Exception as e: print("$name ERR", e, open("$O/bench_$name.err").read()[-500:])
P
}
python -c "
import sys; sys.path.insert(0,'$R')
from rfdnet_amd import sharding; c=sharding.pin_cpus_for_rank(0); print('pin', c and (len(c), c[0], c[-1]))"
for i in 1 2 3; do
ENVX="A=1"; run r03_$i $R/.r03tree --steps 8 --warmup 3
ENVX="A=1"; run r04_$i $R --steps 8 --warmup 3
ENVX="RFD_PIN_NUMA=0"; run r04_nopin_$i $R --steps 8 --warmup 3
done
ENVX="A=1"; run r03_20 $R/.r03tree --steps 20 --warmup 5
ENVX="A=1"; run r04_20 $R --steps 20 --warmup 5
ENVX="A=1"; run r03_m128 $R/.r03tree --config mise128 --steps 4 --warmup 2
ENVX="A=1"; run r04_m128 $R --config mise128 --steps 4 --warmup 2
ENVX="A=1"; run r03_dense32 $R/.r03tree --config dense32 --steps 4 --warmup 2
ENVX="A=1"; run r04_dense32 $R --config dense32 --steps 4 --warmup 2
