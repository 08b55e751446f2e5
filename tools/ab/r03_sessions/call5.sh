set -x
R=$PWD
mkdir -p gpurun_out/c5
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/c5/pytest.txt 2>&1
python bench.py --stats-out gpurun_out/c5/scenes.json > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c5/kt -- python $R/bench.py --no-cpu-baseline --steps 3 --warmup 2 > $R/gpurun_out/c5/bench_profiled.json 2> $R/gpurun_out/c5/kt.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c5/pmc/fetch -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/c5/pmc_fetch.json 2> $R/gpurun_out/c5/pmc_fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c5/pmc/write -- python $R/bench.py --no-cpu-baseline --no-latency --in-flight 1 --steps 2 --warmup 1 > $R/gpurun_out/c5/pmc_write.json 2> $R/gpurun_out/c5/pmc_write.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/c5/pmc/gfetch -- python $R/tools/group_only.py > /dev/null 2> $R/gpurun_out/c5/g1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/c5/pmc/gwrite -- python $R/tools/group_only.py > /dev/null 2> $R/gpurun_out/c5/g2.err
cd $R
ls gpurun_out/c5/kt | head; find gpurun_out/c5/kt -name "*.db" | head -2
DB=$(find gpurun_out/c5/kt -name "*.db" | head -1)
python tools/rocpd_stats.py $DB 45 > gpurun_out/c5/kernel_trace.txt 2>&1
python tools/pmc_traffic.py --group gpurun_out/c5/pmc r03 > gpurun_out/c5/group_pmc.txt 2>&1
cat gpurun_out/c5/group_pmc.txt
# keep only the csv summaries of the PMC passes (size)
find gpurun_out/c5/pmc -name "*.csv" | head -20; du -sh gpurun_out/c5
tail -4 gpurun_out/c5/pytest.txt; cut -c1-300 gpurun_out/c5/bench.json; head -8 gpurun_out/c5/kernel_trace.txt
