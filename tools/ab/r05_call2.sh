# round 5, GPU call 2: FPS abort tests (CU-holding hook), clean single-scene traces (mise128, demo), matrix-pipe-busy counters
mkdir -p gpurun_out/r5c2
O=$PWD/gpurun_out/r5c2
R=$PWD
timeout 600 python -m pytest tests/test_gpu_fps_abort.py tests/test_gpu_ops.py tests/test_gpu_gemm.py tests/test_gpu_decoder.py -m gpu -x -q -p no:cacheprovider > $O/pytest.txt 2>&1; tail -4 $O/pytest.txt
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $O/ss -o ss -- python $R/bench.py --config mise128 --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 3 --warmup 1 > $O/m128_ss.json 2> $O/m128_ss.err
DB=$(find $O/ss -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB --last-scene > $O/m128_single_scene_kernel_trace.txt 2>&1; head -48 $O/m128_single_scene_kernel_trace.txt | cut -c1-175
rm -rf $O/ss
timeout 300 rocprofv3 --kernel-trace -d $O/sd -o sd -- python $R/bench.py --config demo --no-cpu-baseline --no-extras --no-latency --in-flight 1 --steps 4 --warmup 2 > $O/demo_ss.json 2> $O/demo_ss.err
DB=$(find $O/sd -name "*.db" | head -1); python $R/tools/rocpd_stats.py $DB --last-scene > $O/demo_single_scene_kernel_trace.txt 2>&1; head -40 $O/demo_single_scene_kernel_trace.txt | cut -c1-175
rm -rf $O/sd
n=0
for cs in "SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT"; do
n=$((n+1))
timeout 200 rocprofv3 --kernel-trace --pmc $cs --output-format csv -d $O/sq3/$n -- python $R/tools/dec_only.py 3 > /dev/null 2> $O/sq3_$n.err
done
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/sq1/1 -- python $R/tools/dec_only.py 1 > /dev/null 2> $O/sq1_1.err
cd $R
python tools/pmc_sq.py $O/sq3 "occ_decode8_kernelILi3E" > $O/decoder8_pmc_f16x3.txt 2>&1; cat $O/decoder8_pmc_f16x3.txt
python tools/pmc_sq.py $O/sq1 "occ_decode8_kernelILi1E" > $O/decoder8_pmc_f16x1.txt 2>&1; cat $O/decoder8_pmc_f16x1.txt
rm -rf $O/sq3 $O/sq1
