"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): python tools/kstats.py <results.db> [top]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = ("select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start), sum(d.end-d.start), s.arch_vgpr_count, s.private_segment_size "
     "from %s d join %s s on d.kernel_id=s.id group by s.kernel_name order by 5 desc limit %d" % (kd, ks, int(sys.argv[2]) if len(sys.argv) > 2 else 16))
print("kernel,calls,avg_us,min_us,total_ms,vgprs,scratch")
for r in c.execute(q):
    name = r[0].replace("_ZN2fg12_GLOBAL__N_1", "")
    print("%s,%d,%.1f,%.1f,%.2f,%s,%s" % (name[:90], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e6, r[5], r[6]))
