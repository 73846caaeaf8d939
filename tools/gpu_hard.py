"""Suffix sorter on long-common-prefix inputs (developer tool, run on the GPU box): per input the time of the BWT forward stage of one
encode, the number of doubling rounds, and whether the stream equals the reference's digest in tests/golden/golden_full.json."""
import hashlib, importlib, json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib, vectors
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
fr = importlib.import_module("kanzi_amd.framing")
recs = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "golden_full.json")))
ctx = hipapi.Context(0)
only = sys.argv[1].split(",") if len(sys.argv) > 1 else None
for rec in recs:
    name = str(rec["config"])
    if not name.startswith("hard:") or (only and name[5:] not in only):
        continue
    d = vectors.make(tuple(rec["input"]))
    p = ctx.params(rec["transform"], rec["entropy"], rec["block"])
    hdr, hb = fr.make_header(p.entropy_type, p.transform_type, rec["block"], 0, rec["orig_size"])
    cap = ctx.encode_bound(p, len(d)) + 64
    d_in, d_out = ctx.malloc(len(d) + 64), ctx.malloc(cap)
    ctx.h2d(d_in, d)
    ctx.encode_blocks(p, d_in, len(d), d_out, cap, prologue=hdr, prologue_bits=hb)     # warm-up (workspaces)
    ctx.set_profiling(True)
    t0 = time.time()
    bits = ctx.encode_blocks(p, d_in, len(d), d_out, cap, prologue=hdr, prologue_bits=hb)
    wall = (time.time() - t0) * 1e3
    kt = ctx.kernel_times()
    ctx.set_profiling(False)
    out = ctx.d2h(d_out, (bits + 7) // 8)
    ok = len(out) == rec["out"]["len"] and hashlib.md5(out).hexdigest() == rec["out"]["md5"]
    bwt = sum(ms for n, ms, l in kt if n.startswith("k_bwt_f"))
    rounds = sum(l for n, ms, l in kt if n == "k_bwt_f_round")
    top = sorted([(ms, n, l) for n, ms, l in kt if n.startswith("k_bwt_f")], reverse=True)[:5]
    print("%-18s %9d B  bwt_forward %8.2f ms  (%6.1f MB/s)  encode wall %8.2f ms  rounds %2d  reference stream: %s" %
          (name, len(d), bwt, len(d) / bwt / 1e3 if bwt else 0, wall, rounds, "identical" if ok else "DIFFERENT"), flush=True)
    print("    " + ", ".join("%s %.2f ms x%d" % (n, ms, l) for ms, n, l in top), flush=True)
    ctx.free(d_in); ctx.free(d_out)
