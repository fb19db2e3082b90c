# Diagnostic build with in-kernel s_memtime stamps (metric variant only) + phase report.
set -e
cd $(dirname $0)/../vae_training_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DVAEK_STAMPS -DVAEK_FUSED_ONLY_M"
rm -rf /tmp/vaek_st && mkdir -p /tmp/vaek_st          # a private directory: every object of csrc/
for f in $(ls *.hip | sed "s/\.hip$//"); do /opt/rocm/bin/hipcc $F -c $f.hip -o /tmp/vaek_st/$f.o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libvaek_stamps.so /tmp/vaek_st/*.o
cd ../.. && python3 tools/stamps.py "$@"
