# HBM-side traffic per launch of the benchmark's kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes over
# one frame of the default benchmark workload.  Writes gpurun_out/pmc_traffic_bench.json.
cd $GRAFT_REPO_ROOT && export TMPDIR=/tmp
rm -rf gpurun_out/pmc_bfetch gpurun_out/pmc_bwrite
timeout -s KILL 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_bfetch -o p -- python3 bench.py --steps 1 --warmup 0 --timed-frames-only > gpurun_out/pmc_bfetch.log 2>&1 || { echo fetch pass failed; tail -3 gpurun_out/pmc_bfetch.log; exit 1; }
timeout -s KILL 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_bwrite -o p -- python3 bench.py --steps 1 --warmup 0 --timed-frames-only > gpurun_out/pmc_bwrite.log 2>&1 || { echo write pass failed; tail -3 gpurun_out/pmc_bwrite.log; exit 1; }
python3 - <<'PY'
import csv, collections, json, re, os
out = {}
for name, path in (("FETCH_SIZE", "gpurun_out/pmc_bfetch/p_counter_collection.csv"), ("WRITE_SIZE", "gpurun_out/pmc_bwrite/p_counter_collection.csv")):
    acc = collections.defaultdict(float); n = collections.Counter()
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] != name:
                continue
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            m = re.match(r"([\w:]+(<[^>]*>)?)", k); k = m.group(1) if m else k
            acc[k] += float(row["Counter_Value"]); n[k] += 1
    for k in acc:
        out.setdefault(k, {})[name + "_KB_per_launch"] = acc[k] / n[k]
        out[k]["launches"] = n[k]
    os.remove(path)
for k, v in out.items():
    # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes, so reads count double; WRITE_SIZE is exact
    v["traffic_bytes_per_launch"] = 1024.0 * (2.0 * v.get("FETCH_SIZE_KB_per_launch", 0.0) + v.get("WRITE_SIZE_KB_per_launch", 0.0))
json.dump({"command": "python3 bench.py --steps 1 --warmup 0 --timed-frames-only", "workload": "pt1m_4k_256spp", "kernels": out},
          open("gpurun_out/pmc_traffic_bench.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
