"""Median / minimum duration, launch count of every kernel in the second half of a rocprofv3 --kernel-trace run:
    rocprofv3 --kernel-trace -d DIR -o t --output-format csv -- python3 <script>;  python3 tools/trace_kernels.py DIR"""
import csv, sys, glob, collections, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]
acc = collections.defaultdict(list)
for r in rows:
    k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    k = re.sub(r"^void ", "", k).split("(")[0][:64]
    acc[(k, r.get("Grid_Size", "?"), r.get("Workgroup_Size", "?"), r.get("LDS_Block_Size", "?"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for (k, g, w, l), v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 5: continue
    v.sort()
    print("%7.1f us med %7.1f min  x%4d  grid %8s wg %5s lds %6s  %s" % (v[len(v) // 2] / 1e3, v[0] / 1e3, len(v), g, w, l, k))
