set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmc2.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc1","pmc2"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:40]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in agg.items():
            if "fused" not in k: continue
            print(d,k,{c: round(sum(x)/len(x)) for c,x in v.items()})
PY
