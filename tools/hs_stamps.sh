# Diagnostic build of libvaek with s_memtime stamps in the bf16-storage GEMM main loop; prints the phase shares.
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/hsst && for f in $(ls *.hip | sed "s/\.hip$//"); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_HS_STAMPS -c $f.hip -o /tmp/hsst/$f.o &
done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/hsst/libvaek.so /tmp/hsst/*.o
cd $GRAFT_REPO_ROOT
VAEK_LIB_PATH=/tmp/hsst/libvaek.so python3 tools/hs_stamps.py "$@"
