"""SRT inverse of one block of BWT(text) with its wall time (developer probe: run under rocprofv3 --pmc for instruction counts)."""
import sys, os, time, importlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); import knzlib
knzlib.load_pkg(); hipapi = importlib.import_module("kanzi_amd.hipapi")
ctx = hipapi.Context(0)
corpus = importlib.import_module("kanzi_amd.corpus")
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 4
d = corpus.text(MB << 20, 1)
# (both forward transforms on the device: the oracle's suffix sort takes minutes at 32 MiB)
ok, bw = ctx.transform_forward("BWT", d, len(d) + 64, "FPAQ")
ok, sr = ctx.transform_forward("SRT", bw, len(bw) + 2048, "FPAQ")
import numpy as np
a = np.frombuffer(bw, dtype=np.uint8); print("runs:", int((a[1:] != a[:-1]).sum()) + 1)
for rep in range(2):
    t0 = time.time(); ok2, back = ctx.transform_inverse("SRT", sr, len(bw)); t1 = time.time()
    print(ok2, back == bw, round(t1 - t0, 4))
ctx.set_profiling(True); ctx.transform_inverse("SRT", sr, len(bw))
print([(n, round(ms, 2)) for n, ms, _ in ctx.kernel_times() if ms > 0.05])
