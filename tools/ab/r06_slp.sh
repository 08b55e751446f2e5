# round 6: the library built with -fno-slp-vectorize (shipped) against the same sources without it (rfdnet_amd/lib/variants/librfd_slp.so,
# the round 1-5 flags): tail-kernel bit identity, the frozen decoder's rate, the encoder GEMMs, the headline -- one box, interleaved.
#   bash tools/ab/r06_slp.sh   -> gpurun_out/r6slp/*
O=$PWD/gpurun_out/r6slp; mkdir -p $O; V=$PWD/rfdnet_amd/lib/variants/librfd_slp.so
# the control library: today's sources with the flags of rounds 1-5 (built here when it did not travel with the snapshot)
if [ ! -f $V ]; then mkdir -p $(dirname $V); /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -fvisibility=hidden -Iinclude -o $V rfdnet_amd/csrc/*.hip || exit 1; fi
[ -x tools/hazard/pk_f32_under_mfma ] || bash tools/hazard/build.sh || exit 1
python tools/hazard/tail_vs_main.py 2>&1 | grep "mode\|tail\|more\|determinism" > $O/diag_new.txt; cat $O/diag_new.txt
RFD_HIP_LIB=$V python tools/hazard/tail_vs_main.py 2>&1 | grep "mode\|tail\|more\|determinism" > $O/diag_slp.txt; cat $O/diag_slp.txt
tools/hazard/pk_f32_under_mfma 50000 > $O/pk_f32_under_mfma.txt 2>&1
tools/hazard/load_valu_under_mfma 20000 > $O/load_valu_under_mfma.txt 2>&1
tools/hazard/mfma_late_read 2000 > $O/mfma_late_read.txt 2>&1
tools/hazard/mfma_raw_valu 4000 > $O/mfma_raw_valu.txt 2>&1
for r in 1 2; do
  for lib in new slp; do
    if [ $lib = slp ]; then export RFD_HIP_LIB=$V; else unset RFD_HIP_LIB; fi
    timeout 300 python bench.py --config stress --steps 3 --warmup 1 --no-cpu-baseline > $O/stress_${lib}_$r.json 2> $O/stress_${lib}_$r.err
    timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/headline_${lib}_$r.json 2> $O/headline_${lib}_$r.err
    timeout 200 python tools/gemm_frag_bench.py > $O/gemm_${lib}_$r.txt 2>&1
  done
done
unset RFD_HIP_LIB
python - <<P
import json,glob
for f in sorted(glob.glob("$O/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], "%.4g %s frac %.4f" % (d["value"], d["unit"], d["roofline"]["frac"]))
    except Exception as e: print(f, "ERR", e)
P
grep -h "total\|ten GEMMs\|sum" $O/gemm_*.txt | head -8
