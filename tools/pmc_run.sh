# Counter passes (rocprofv3 --pmc, one group per run, --kernel-trace only) over any python command of this repo.
#   bash tools/pmc_run.sh <tag> <kernel-name filter (python substring)> <script and args ...>
# Writes gpurun_out/pmc_<tag>.txt: per kernel and counter, the average over its launches.
set -e
TAG=$1; FILTER=$2; shift 2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_${TAG}_$i -- python3 $R/$@ > $R/gpurun_out/pmc_${TAG}_$i.log 2>&1 || echo "pass $i ($grp) failed"
done
cd $R
python3 - "$TAG" "$FILTER" <<'PY'
import csv, glob, collections, sys
tag, flt = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if flt in k:
            agg[k[:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for f in glob.glob(f"gpurun_out/pmc_{tag}_1/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            dur[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
with open(f"gpurun_out/pmc_{tag}.txt", "w") as out:
    for k, v in sorted(agg.items()):
        line = f"{k}\n   launches {max(len(x) for x in v.values())}  avg_us(under pmc) {sum(dur[k]) / max(len(dur[k]), 1):.1f}\n   " + \
               "  ".join(f"{c}={sum(x) / len(x):.0f}" for c, x in sorted(v.items()))
        print(line); out.write(line + "\n")
PY
