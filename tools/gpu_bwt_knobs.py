"""Suffix sorter timing under different knob settings (developer tool): BWT forward stage of one encode, HIP-event time per kernel.
usage: gpu_bwt_knobs.py corpus[,corpus...] "knob=value,knob=value;knob=value;..." [bytes]"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
c = knzlib.corpus()
kinds = sys.argv[1].split(",")
settings = sys.argv[2].split(";") if len(sys.argv) > 2 else [""]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 211957760
ctx = hipapi.Context(0)
L = hipapi.lib()
L.knz_hip_tune(b"bwt_split", 1)
for kind in kinds:
    d = {"mixed": lambda: c.mixed(n, 2), "text": lambda: c.text(n, 1), "repeats": lambda: c.repeats(n, 3), "local": lambda: c.local(n)[0]}[kind]()
    p = ctx.params("BWT", "NONE", 8 << 20)
    cap = ctx.encode_bound(p, len(d)) + 64
    d_in, d_out = ctx.malloc(len(d) + 64), ctx.malloc(cap)
    ctx.h2d(d_in, d)
    ref = None
    for st in settings:
        names = []
        for kv in [x for x in st.split(",") if x]:
            k, v = kv.split("=")
            assert L.knz_hip_tune(k.encode(), int(v)) == 0, k
            names.append(k)
        ctx.encode_blocks(p, d_in, len(d), d_out, cap)
        ctx.set_profiling(True)
        bits = ctx.encode_blocks(p, d_in, len(d), d_out, cap)
        kt = ctx.kernel_times()
        ctx.set_profiling(False)
        out = ctx.d2h(d_out, (bits + 7) // 8)
        if ref is None:
            ref = out
        bwt = sum(ms for nm, ms, l in kt if nm.startswith("k_bwt_f"))
        rounds = sum(l for nm, ms, l in kt if nm == "k_bwt_f_round")
        top = sorted([(ms, nm, l) for nm, ms, l in kt if nm.startswith("k_bwt_f")], reverse=True)[:9]
        print("%-8s %-44s bwt_forward %7.2f ms rounds %2d %s | %s" % (kind, st or "(defaults)", bwt, rounds, "same" if out == ref else "DIFFERENT OUTPUT",
              ", ".join("%s %.2f" % (nm[8:], ms) for ms, nm, l in top)), flush=True)
        for k in names:
            L.knz_hip_tune(k.encode(), 1 if k == "bwt_link" else 0)      # (back to the default)
    ctx.free(d_in); ctx.free(d_out)
