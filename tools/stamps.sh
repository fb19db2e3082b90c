# Diagnostic build with in-kernel s_memtime stamps (metric variant only) + phase report.
set -e
cd $(dirname $0)/../vae_training_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_STAMPS -DVAEK_FUSED_ONLY_M"
for f in $(ls *.hip | sed "s/\.hip$//"); do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/st_$f.o; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvaek_stamps.so /tmp/st_api.o /tmp/st_gemm_f32.o /tmp/st_gemm_bf16.o /tmp/st_elbo.o /tmp/st_fused_small.o /tmp/st_fused_mfma.o /tmp/st_comm.o /tmp/st_rng.o /tmp/st_microbench.o
cd ../.. && python3 tools/stamps.py "$@"
