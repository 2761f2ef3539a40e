"""Per-kernel mean of rocprofv3 counter values: python tools/pmc_summary.py gpurun_out/pmc_<tag> [...]"""
import csv, glob, sys, collections, re

def short(name):
    m = re.search(r"(k_[a-z0-9_]+)", name)
    return m.group(1) if m else name[:40]

for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in sorted(acc):
            print(k, {c: round(sum(v) / len(v), 1) for c, v in acc[k].items()}, "launches", len(next(iter(acc[k].values()))))
