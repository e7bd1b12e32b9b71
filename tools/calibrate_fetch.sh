# FETCH_SIZE against known byte counts (tools/calibrate_fetch.hip) for tables of 2 GiB (every record an L2 and
# Infinity Cache miss), 128 MiB and 64 MiB (the size of the benchmark's tree: L2 misses served by the Infinity Cache).
# Writes gpurun_out/fetch_calibration.json.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
mkdir -p gpurun_out/calib
for mib in 2048 128 64; do
  rm -rf gpurun_out/calib/pmc_$mib
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/pmc_$mib -o p -- ./tools/calibrate_fetch $mib > gpurun_out/calib/run_$mib.log 2>&1 || { echo "pass $mib failed"; tail -3 gpurun_out/calib/run_$mib.log; exit 1; }
  timeout -k 10 120 ./tools/calibrate_fetch $mib > gpurun_out/calib/time_$mib.log 2>&1 || exit 1
done
python3 - <<'PY'
import csv, json, collections
doc = {}
for mib in (2048, 128, 64):
    req = collections.defaultdict(list)
    for line in open(f"gpurun_out/calib/run_{mib}.log"):
        f = line.split()
        if len(f) >= 9 and f[1] == "table_MiB":
            req[f[0]].append(float(f[4]))
    rate = collections.defaultdict(list)
    for line in open(f"gpurun_out/calib/time_{mib}.log"):
        f = line.split()
        if len(f) >= 9 and f[1] == "table_MiB":
            rate[f[0]].append(float(f[8]))
    fetch = collections.defaultdict(list)
    with open(f"gpurun_out/calib/pmc_{mib}/p_counter_collection.csv") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] == "FETCH_SIZE":
                k = row["Kernel_Name"].split("(")[0].replace("void ", "")
                fetch[k].append(float(row["Counter_Value"]) * 1024.0)
    doc[str(mib)] = {k: {"requested_bytes": req[k][-1], "FETCH_SIZE_bytes": fetch[k][-1], "requested_over_FETCH_SIZE": req[k][-1] / fetch[k][-1],
                         "GBps_unprofiled": max(rate[k])} for k in req if k in fetch}
json.dump(doc, open("gpurun_out/fetch_calibration.json", "w"), indent=1)
print(json.dumps(doc, indent=1))
PY
