#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; each with --kernel-trace only) of `bench.py` into HBM traffic per
pipeline STAGE and per encode+decode step, and store it in profiles/pmc_traffic.json (read by bench.py for roofline.traffic).

usage: pmc_stage_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <configN> <out.json> [txt]

Units and corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950
FETCH_SIZE tallies the 128-B requests of coalesced reads at 64 B, so it is doubled -- for streaming kernels. The calibration
of round 3 (tools/membench.hip under rocprofv3, profiles/r03_pmc_calibration.txt) on 222 M random 4-byte gathers and dependent
8-byte walks shows 62-64 B of FETCH_SIZE per access, i.e. one 64-B request each, which is also what the measured rate (55 G/s
= 3.5 TB/s) allows; doubling that would claim 7 TB/s of reads. Kernels that are dominated by such gathers (GATHER below) are
therefore NOT doubled. WRITE_SIZE needs no correction: exact for coalesced stores, 32 B per random 4-byte store.
A dispatch belongs to the stage of the bracket it falls in (k_bwt_bases .. k_bwt_f_emit = bwt_forward, k_bwt_i_header ..
k_bwt_i_place = bwt_inverse: the scan / sort primitives in between carry generic names), otherwise to the stage its own name
says. Per step = total / number of encodes seen (dispatches of the first kernel of an encode: k_bwt_bases, k_ans0_stats, k_huff_encode, k_lz_keys or k_ans1_hist)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import stage_of  # noqa: E402

OPEN = {"k_bwt_bases": "bwt_forward", "k_bwt_i_header": "bwt_inverse"}
CLOSE = {"k_bwt_f_emit": "bwt_forward", "k_bwt_i_place": "bwt_inverse"}
# read traffic of these kernels is random 4/8-byte gathers: FETCH_SIZE is not doubled (see above)
GATHER = ("k_bwt_f_gather_small", "k_bwt_f_small_fused", "k_bwt_f_gather_desc", "k_bwt_i_walk", "k_bwt_f_large_keys", "k_bwt_f_run_table", "k_bwt_i_jump", "k_bwt_f_emit")


def read_factor(name):
    return 1.0 if name.startswith(GATHER) else 2.0


def short(name):
    n = name.split("(")[0].replace("void ", "")
    n = n.replace("knz::", "")
    return n.split("<")[0]


def per_stage(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
    tot = collections.defaultdict(float)
    kern = collections.defaultdict(float)
    calls = collections.Counter()
    cur = None
    for r in rows:
        n = short(r["Kernel_Name"])
        if n in OPEN:
            cur = OPEN[n]
        st = cur if cur else stage_of(n, "step")
        v = float(r["Counter_Value"]) * (read_factor(n) if counter == "FETCH_SIZE" else 1.0)
        tot[st] += v
        kern[(st, n[:40])] += v
        calls[n] += 1
        if n in CLOSE:
            cur = None
    return tot, kern, calls


def main():
    fpath, wpath, cfg, out = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    ft, fk, calls = per_stage(fpath, "FETCH_SIZE")
    wt, wk, _ = per_stage(wpath, "WRITE_SIZE")
    n_enc = calls.get("k_bwt_bases") or calls.get("k_ans0_stats") or calls.get("k_huff_encode") or calls.get("k_lz_keys") or calls.get("k_ans1_hist") or 1
    stages = {}
    for st in sorted(set(ft) | set(wt)):
        stages[st] = int((ft.get(st, 0.0) * 1024 + wt.get(st, 0.0) * 1024) / n_enc)
    try:
        full = json.load(open(out))
    except Exception:
        full = {}
    n_bytes = int(os.environ.get("KNZ_PMC_NBYTES", "211957760"))
    full["config" + cfg] = {"n_bytes": n_bytes, "stages": stages, "encodes_seen": n_enc,
                            "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950) except for gather kernels (x1, calibrated), KiB -> bytes, per step",
                            **{k: v for k, v in full.get("config" + cfg, {}).items() if k not in ("n_bytes", "stages", "encodes_seen", "source")}}
    json.dump(full, open(out, "w"), indent=1, sort_keys=True)
    if len(sys.argv) > 5:
        with open(sys.argv[5], "w") as t:
            t.write("# HBM traffic per stage and per step (one encode + one decode of the corpus) from rocprofv3 --pmc, config %s\n" % cfg)
            t.write("# FETCH_SIZE x2 (gfx950 correction; x1 for gather kernels, calibrated) and WRITE_SIZE, KiB -> bytes; %d encodes seen in the run\n" % n_enc)
            t.write("%-18s %16s %16s\n" % ("stage", "read bytes", "write bytes"))
            for st in sorted(stages, key=lambda s: -stages[s]):
                t.write("%-18s %16d %16d\n" % (st, ft.get(st, 0.0) * 1024 / n_enc, wt.get(st, 0.0) * 1024 / n_enc))
            t.write("\n# largest kernels (corrected read + write, bytes per step)\n")
            allk = collections.defaultdict(float)
            for k, v in fk.items():
                allk[k] += v * 1024 / n_enc
            for k, v in wk.items():
                allk[k] += v * 1024 / n_enc
            for (st, n), v in sorted(allk.items(), key=lambda kv: -kv[1])[:25]:
                t.write("%-18s %-42s %16d\n" % (st, n, v))


if __name__ == "__main__":
    main()
