#!/usr/bin/env python3
"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_traffic.json.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <n_bytes> <configN> <out.json> [txt]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in
KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled.  WRITE_SIZE is taken as is
(it reproduces the known byte counts of k_ans0_decode's output and k_assemble's stream exactly).
Values are the mean per launch of each kernel, in bytes.
"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and r["Kernel_Name"].startswith("knz::"):
            d[r["Kernel_Name"].split("(")[0].replace("knz::", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in d.items()}


def main():
    fpath, wpath, n, key, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    f = per_kernel(fpath, "FETCH_SIZE")
    w = per_kernel(wpath, "WRITE_SIZE")
    res = {"n_bytes": n, "note": "bytes per launch; FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes"}
    rows = []
    for k in sorted(set(f) | set(w)):
        rd = f.get(k, 0.0) * 2 * 1024
        wr = w.get(k, 0.0) * 1024
        res[k] = int(rd + wr)
        rows.append((k, rd, wr))
    try:
        full = json.load(open(out))
    except Exception:
        full = {}
    full[key] = res
    json.dump(full, open(out, "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 6:
        with open(sys.argv[6], "w") as t:
            t.write("# HBM traffic per launch from rocprofv3 --pmc (separate FETCH_SIZE / WRITE_SIZE passes)\n")
            t.write("# input bytes per launch: %d\n" % n)
            t.write("%-28s %16s %16s\n" % ("kernel", "read bytes (x2)", "write bytes"))
            for k, rd, wr in sorted(rows, key=lambda r: -(r[1] + r[2])):
                t.write("%-28s %16d %16d\n" % (k, rd, wr))


if __name__ == "__main__":
    main()
