#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (kernel-trace) into a per-kernel table:
   python tools/rocprof_summary.py gpurun_out/prof/x_results.db > profiles/rNN_kernel_stats.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute("select %s, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by %s order by 3 desc" % (name_col, name_col)).fetchall()
tot = sum(r[2] for r in rows) or 1
print("%-60s %8s %14s %12s %12s %12s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
for r in rows:
    print("%-60s %8d %14d %12.0f %12d %12d %6.2f%%" % (r[0][:60], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
