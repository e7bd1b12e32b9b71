"""Sums rocprofv3 counter_collection CSV files per (kernel, counter)."""
import csv, sys, collections, re
for path in sys.argv[1:]:
    acc = collections.defaultdict(float)
    n = collections.Counter()
    with open(path) as f:
        for row in csv.DictReader(f):
            k = row["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
            m = re.match(r"([\w:]+(<[^>]*>)?)", k)
            k = m.group(1) if m else k
            acc[(k, row["Counter_Name"])] += float(row["Counter_Value"])
            n[k] += 1
    kernels = sorted({k for k, _ in acc})
    counters = sorted({c for _, c in acc})
    print(path)
    for k in kernels:
        print(" ", k)
        for c in counters:
            if (k, c) in acc:
                print(f"    {c:32s} {acc[(k, c)]:.4g}")
    import os
    os.remove(path)   # large; the sums above are what is kept
