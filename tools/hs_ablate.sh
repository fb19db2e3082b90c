# Diagnostic builds of the persistent bf16-storage GEMM: 1 = no MFMAs, 2 = no output stores, 4 = no operand loads (bit mask).
# Prints the forward / dX kernel times of each build (outputs are wrong by construction; only the clock matters).
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
mkdir -p /tmp/hsab_common
for f in $(ls *.hip | sed "s/\.hip$//" | grep -v "^gemm_bf16s$"); do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -c $f.hip -o /tmp/hsab_common/$f.o &
done; wait
for ab in "$@"; do
  mkdir -p /tmp/hsab$ab
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_HS_ABLATE=$ab -c gemm_bf16s.hip -o /tmp/hsab$ab/gemm_bf16s.o &
done; wait
for ab in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/hsab$ab/libvaek.so /tmp/hsab_common/*.o /tmp/hsab$ab/gemm_bf16s.o; done
cd $GRAFT_REPO_ROOT
for ab in "$@"; do
  echo "== ablate mask $ab"
  VAEK_LIB_PATH=/tmp/hsab$ab/libvaek.so python3 tools/hs_tune.py 65536 ${HS_V:-11} 0 2>&1 | tail -2
done
