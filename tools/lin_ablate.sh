# Diagnostic builds of linear_moments.hip with parts of the streamers removed (-DVAEK_LIN_ABL=mask: 1 pieces issued in front of the
# products, 2 no MFMAs, 4 no image store, 8 no LDS-DMA), streamers-only timing of each (tools/lin_roles.sh, VAEK_LIN_ROLES=1).
set -e
cd $GRAFT_REPO_ROOT/vae_training_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc"
mkdir -p /tmp/linabl
for f in $(ls *.hip | sed "s/\.hip$//" | grep -v linear_moments); do /opt/rocm/bin/hipcc $FL -c $f.hip -o /tmp/linabl/$f.o & done
for m in ${MASKS:-0 1 2 4 8 10 14}; do /opt/rocm/bin/hipcc $FL -DVAEK_LIN_ABL=$m -c linear_moments.hip -o /tmp/linabl/lm_$m.o & done
wait
cd $GRAFT_REPO_ROOT
for m in ${MASKS:-0 1 2 4 8 10 14}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/linabl/libvaek_$m.so $(ls /tmp/linabl/*.o | grep -v "/lm_") /tmp/linabl/lm_$m.o
  echo "ablation mask $m"
  ROLES_LIST=1 VAEK_LIB_PATH=/tmp/linabl/libvaek_$m.so bash tools/lin_roles.sh 2>&1 | grep "roles 1"
done
