"""One encode's BWT forward launches in order with their durations, from a rocprofv3 --kernel-trace CSV (developer tool).
usage: rocprof_trace_list.py <kernel_trace.csv> [occurrence]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
occ = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("knz::", "")
    r["Kernel_Name"] = n.split("(")[0]
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_bwt_bases")]
ends = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("k_bwt_f_emit")]
a, b = starts[occ], ends[occ]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    nm = r["Kernel_Name"][:44]
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, nm))
    prev_end = e
print("total %.1f us" % ((prev_end - t0) / 1e3))
