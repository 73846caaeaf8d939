#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE of tools/membench.hip's access patterns (known byte counts) -> bytes per access as the counters report
them: the calibration behind the per-kernel corrections of tools/pmc_stage_summary.py.
usage: pmc_calibration.py <fetch counter_collection.csv> <write counter_collection.csv> <membench stdout>"""
import collections
import csv
import sys

N = 222298112          # accesses per launch (membench)
SETS = [1, 8, 32, 64, 128, 192, 256, 512, 1024]


def load(path, counter):
    out = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]) * 1024)
    return out


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
print("# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) on tools/bin/membench pmc: %d accesses per launch, two launches per shape" % N)
print("# counters as reported (KiB -> bytes), NO correction applied")
print("copy of 889 MB (16 B per lane):        FETCH %.0f MB (true 889 MB: x2 needed)   WRITE %.0f MB (true 889 MB: exact)" % (f["k_copy"][0] / 1e6, w["k_copy"][0] / 1e6))
g, s = f["k_gather4<4>"], w["k_scatter4<4>"]
sr = f["k_scatter4<4>"]
print("coalesced 4 B per lane read of 889 MB: FETCH %.0f MB (x2 needed as well)" % (sr[0] / 1e6))
print("%-10s %28s %28s" % ("set (MB)", "random 4 B gather: FETCH B/access", "random 4 B scatter: WRITE B/access"))
for i, ws in enumerate(SETS):
    print("%-10d %28.1f %28.1f" % (ws, g[2 * i] / N, s[2 * i] / N))
wk = f["k_walk8"]
print("dependent 8 B record walk (222 M hops), sets 32 MB .. 1.7 GB: FETCH B/hop " + " ".join("%.1f" % (wk[2 * i] / N) for i in range(6)))
print("# => a random gather is ONE 64-B request in FETCH_SIZE (no doubling), a random 4-byte store is 32 B in WRITE_SIZE; streaming reads are")
print("#    reported at half their bytes (the guide's x2), streaming writes exactly.")
print()
print(open(sys.argv[3]).read())
