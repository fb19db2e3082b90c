# HBM traffic of the bench kernels from the L2 memory-side counters (separate --pmc passes with --kernel-trace only, as
# MI355X_MICROARCH.md prescribes; FETCH_SIZE / WRITE_SIZE count KiB on rocprofv3, and on gfx950 FETCH_SIZE reports HALF of the
# fetched bytes).   bash tools/pmc_traffic.sh <workload> <f32|bf16> [more bench.py flags]
# Writes gpurun_out/r03_pmc_traffic_<workload>[_bf16].json in the form bench.py's roofline leg reads from profiles/.
set -e
WL=${1:-M}; DT=${2:-f32}; shift 2 || true
STEPS=64; WARM=64
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${WL}$([ "$DT" = bf16 ] && echo _bf16 || true)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${TAG}_$c -- python3 $R/bench.py --workload $WL --dtype $DT --steps $STEPS --warmup $WARM --no-cpu-baseline --no-roofline --graph 0 --repeat 1 --only-main "$@" > $R/gpurun_out/pmc_${TAG}_$c.log 2>&1
done
cd $R
python3 - "$TAG" "$WL" "$DT" "${VAEK_COMMIT:-commit not recorded}" <<'PY'
import csv, glob, collections, json, sys
tag, wl, dt, commit = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
line = [l for l in open(f"gpurun_out/pmc_{tag}_FETCH_SIZE.log") if l.startswith('{"metric')][-1]
nsteps = json.loads(line)["steps_launched_total"]          # train steps bench.py issued on its main entry point (--only-main: no other leg)
LABEL = {"lin_persist_kernel": "lin_moments_persistent", "fused_linear_mfma_kernel": "fused_linear_mfma", "fused_finalize_kernel": "fused_finalize_adam"}
tot = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "vaek" in r["Kernel_Name"] and "microbench" not in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("(")[0]
                tot[name][c] += float(r["Counter_Value"])
                if c == "FETCH_SIZE":
                    cnt[name] += 1
kernels = {}
for name, v in tot.items():
    label = next((lab for key, lab in LABEL.items() if key in name), name)
    per_step = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024 / nsteps
    k = kernels.setdefault(label, {"rocprof_name": name, "launches": 0, "FETCH_SIZE_KiB_per_step": 0.0, "WRITE_SIZE_KiB_per_step": 0.0, "traffic_bytes_per_step": 0})
    k["launches"] += cnt[name]
    k["FETCH_SIZE_KiB_per_step"] += v["FETCH_SIZE"] / nsteps
    k["WRITE_SIZE_KiB_per_step"] += v["WRITE_SIZE"] / nsteps
    k["traffic_bytes_per_step"] += int(per_step)
out = {"note": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_traffic.sh {wl} {dt}): bench.py --workload {wl} --dtype {dt} "
               f"--steps 64 --warmup 64 --graph 0 --repeat 1 --only-main = {nsteps} train steps in all (warm-up, re-warm steps behind the cache sweep, timed); counters summed over "
               f"every launch of a kernel and divided by the steps.  gfx950 correction per MI355X_MICROARCH.md: traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 bytes.",
       "commit": commit, "steps": nsteps, "kernels": kernels,
       "traffic_bytes_per_step_all_kernels": sum(k["traffic_bytes_per_step"] for k in kernels.values())}
json.dump(out, open(f"gpurun_out/r03_pmc_traffic_{tag}.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
