# HBM traffic of the bench kernels from the L2 memory-side counters (separate --pmc passes, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE / WRITE_SIZE are in KiB-units of 1024 B on rocprofv3).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --graph 0 > $R/gpurun_out/pmc_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, collections, json
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{c}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "vaek" in r["Kernel_Name"]:
                agg[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            out[k][c] = sum(v) / len(v)
            out[k]["launches"] = len(v)
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/pmc_traffic_raw.json", "w"), indent=1)
PY
