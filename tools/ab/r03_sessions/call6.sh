R=$PWD
mkdir -p gpurun_out/c6
python tools/ab/dec_ab.py base r0f nt wlo8 wlo4 wlo0 noread nodma neither prio > gpurun_out/c6/ab.txt 2>&1
for v in base r0f wlo0 neither; do
  if [ $v = base ]; then AB_NAME=$v python tools/clock_probe.py | grep PROBE >> gpurun_out/c6/probe.txt 2>&1; else AB_NAME=$v RFD_HIP_LIB=$R/rfdnet_amd/lib/variants/librfd_$v.so python tools/clock_probe.py | grep PROBE >> gpurun_out/c6/probe.txt 2>&1; fi
done
python tools/ab/prio_check.py 13 prio > gpurun_out/c6/prio.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/c6/sq1 -- python $R/tools/dec_only.py 3 > /dev/null 2> $R/gpurun_out/c6/sq1.err
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS --output-format csv -d $R/gpurun_out/c6/sq2 -- python $R/tools/dec_only.py 3 > /dev/null 2> $R/gpurun_out/c6/sq2.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/c6/sq3 -- python $R/tools/dec_only.py 3 > /dev/null 2> $R/gpurun_out/c6/sq3.err
cd $R
RFD_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline --no-latency > gpurun_out/c6/bench_eight_dryrun.json 2> gpurun_out/c6/bench_eight.err
cat gpurun_out/c6/ab.txt; cat gpurun_out/c6/probe.txt; cat gpurun_out/c6/prio.txt; cut -c1-200 gpurun_out/c6/bench_eight_dryrun.json; tail -3 gpurun_out/c6/bench_eight.err
