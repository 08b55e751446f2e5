# stand-alone gfx950 probes of this directory -> executables beside their sources (git-ignored; they travel to the GPU box with the snapshot)
cd "$(dirname "$0")"
for f in pk_f32_under_mfma load_valu_under_mfma mfma_late_read mfma_raw_valu; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w -o $f $f.hip || exit 1
done
