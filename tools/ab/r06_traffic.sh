# round 6: the decoder's HBM traffic inside the benchmark (FETCH_SIZE / WRITE_SIZE in separate passes) -- the part of r06_final.sh that had to be
# repeated after tools/pmc_traffic.py learnt the tail kernel's name.   bash tools/ab/r06_traffic.sh -> gpurun_out/r6traffic/*
R=$PWD; O=$PWD/gpurun_out/r6traffic; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
d=${c%%:*}; ctr=${c##*:}
timeout 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $O/traffic/$d -- python $R/bench.py --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 2 --warmup 1 > $O/bench_pmc_$d.json 2> $O/bench_pmc_$d.err
done
cd $R
python - <<P > $O/npoints.txt
import json
d = json.loads(open("$O/bench_pmc_fetch.json").read().strip().splitlines()[-1])
print(int(3 * d["config"]["queries_per_scene"]))        # the three scenes' nine launches (pmc_traffic.py --first 9)
P
python tools/pmc_traffic.py --first 9 $O/traffic $(cat $O/npoints.txt) r06 > $O/decoder_traffic.txt 2>&1; cat $O/decoder_traffic.txt
cp profiles/decoder_traffic.json $O/decoder_traffic.json
rm -rf $O/traffic
