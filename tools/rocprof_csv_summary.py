#!/usr/bin/env python3
"""Render rocprofv3's <prefix>_kernel_stats.csv (from --kernel-trace --stats --output-format csv) as the short
table kept under profiles/:   python tools/rocprof_csv_summary.py gpurun_out/p5_trace/t_kernel_stats.csv "<command>" """
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
print("# " + (sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]))
print("%-34s %6s %12s %12s %7s" % ("kernel", "calls", "avg_us", "total_ms", "pct"))
for r in rows[:24]:
    name = r["Name"].replace("knz::", "").split("(")[0]
    print("%-34s %6d %12.1f %12.2f %7.4g" % (name[:34], int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                            float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
