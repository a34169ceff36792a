"""Per-kernel averages of a rocprofv3 --pmc pass (rocpd sqlite output): python tools/pmcsum.py <dir> <COUNTER>"""
import collections, glob, os, sqlite3, sys

root, counter = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for db in glob.glob(os.path.join(root, "**", "*_results.db"), recursive=True):
    c = sqlite3.connect(db)
    per_dispatch = collections.defaultdict(float)   # a counter comes as one row per instance: sum them per dispatch
    names = {}
    for name, disp, val in c.execute("select name, dispatch_id, counter_value from pmc_events where counter_name = ?", (counter,)):
        per_dispatch[disp] += val
        names[disp] = name
    for disp, v in per_dispatch.items():
        acc[names[disp]].append(v)
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print(f"{counter} avg {sum(v) / len(v):14.1f}  launches {len(v):3d}  {k[:120]}")
