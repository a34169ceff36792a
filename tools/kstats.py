"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite output): python tools/kstats.py <dir> [header text...]"""
import glob, os, sqlite3, sys

root = sys.argv[1]
dbs = glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True)
if not dbs:
    sys.exit(f"no *_results.db under {root}")
agg = {}
for db in dbs:
    c = sqlite3.connect(db)
    q = ("select s.kernel_name, count(*), sum(d.end - d.start) from rocpd_kernel_dispatch d "
         "join rocpd_info_kernel_symbol s on d.kernel_id = s.id and d.guid = s.guid group by s.kernel_name")
    for name, calls, ns in c.execute(q):
        a = agg.setdefault(name, [0, 0])
        a[0] += calls; a[1] += ns
tot = sum(a[1] for a in agg.values())
if len(sys.argv) > 2:
    print(" ".join(sys.argv[2:]))
for name, (calls, ns) in sorted(agg.items(), key=lambda r: -r[1][1]):
    print(f"{ns / 1e6:9.2f} ms {100 * ns / tot:6.2f}% calls {calls:6d} avg {ns / calls / 1e3:9.1f} us  {name[:110]}")
