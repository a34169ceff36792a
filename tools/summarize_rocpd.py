#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite output (kernel trace) into a per-kernel CSV on stdout.
Usage: python tools/summarize_rocpd.py <dir-with-.db> [top_n]"""
import glob
import sqlite3
import sys

root = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dbs = glob.glob(root + "/**/*.db", recursive=True)
if not dbs:
    sys.exit("no .db under " + root)
con = sqlite3.connect(dbs[0])
rows = con.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"# total kernel time {tot / 1e6:.2f} ms")
print("name,calls,total_ns,avg_ns,min_ns,max_ns,pct")
for r in rows[:top]:
    print(f"\"{r[0][:150]}\",{r[1]},{r[2]},{int(r[3])},{r[4]},{r[5]},{100.0 * r[2] / tot:.2f}")
