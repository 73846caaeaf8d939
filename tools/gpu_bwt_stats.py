"""Per-round populations of the suffix sorter (developer tool): one encode of a corpus with the knob bwt_stats on; the library prints to stderr.
usage: gpu_bwt_stats.py mixed|text|repeats|local [bytes]"""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import knzlib
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
c = knzlib.corpus()
kind = sys.argv[1] if len(sys.argv) > 1 else "mixed"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 211957760
d = {"mixed": lambda: c.mixed(n, 2), "text": lambda: c.text(n, 1), "repeats": lambda: c.repeats(n, 3), "local": lambda: c.local(n)[0]}[kind]()
ctx = hipapi.Context(0)
hipapi.lib().knz_hip_tune(b"bwt_split", 1)
p = ctx.params("BWT", "NONE", 8 << 20)
cap = ctx.encode_bound(p, len(d)) + 64
d_in, d_out = ctx.malloc(len(d) + 64), ctx.malloc(cap)
ctx.h2d(d_in, d)
ctx.encode_blocks(p, d_in, len(d), d_out, cap)
hipapi.lib().knz_hip_tune(b"bwt_stats", 1)
ctx.encode_blocks(p, d_in, len(d), d_out, cap)
