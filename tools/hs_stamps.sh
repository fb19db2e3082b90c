# Diagnostic build of libvaek with s_memtime stamps in the bf16-storage GEMM main loop; prints the phase shares.
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/hsst && for f in api gemm_f32 gemm_bf16 gemm_bf16s gemm_skinny16 elbo fused_small fused_mfma comm rng microbench; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_HS_STAMPS -c $f.hip -o /tmp/hsst/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/hsst/libvaek.so /tmp/hsst/*.o
cd $GRAFT_REPO_ROOT
VAEK_LIB_PATH=/tmp/hsst/libvaek.so python3 tools/hs_stamps.py "$@"
