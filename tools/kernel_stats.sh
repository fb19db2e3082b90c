# rocprofv3 --kernel-trace --stats of one bench.py command; copies the per-kernel summary to gpurun_out/<name>_kernel_stats.csv
#   bash tools/kernel_stats.sh <name> <bench.py flags ...>
set -e
NAME=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ks_$NAME -- python3 $R/bench.py --no-cpu-baseline "$@" > $R/gpurun_out/ks_$NAME.log 2>&1
cd $R
F=$(find gpurun_out/ks_$NAME -name "*kernel_stats.csv" | head -1)
cp "$F" gpurun_out/${NAME}_kernel_stats.csv
grep '^{"metric' gpurun_out/ks_$NAME.log | tail -1 > gpurun_out/${NAME}_line.json
cut -c1-400 gpurun_out/${NAME}_line.json
head -8 gpurun_out/${NAME}_kernel_stats.csv | cut -c1-200
