"""Repeat device round trips and compare every result (developer tool: hunts intermittent faults)."""
import sys, os, time, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import __graft_entry__ as ge
ge.load_package()
hipapi = importlib.import_module("kanzi_amd.hipapi")
corpus = importlib.import_module("kanzi_amd.corpus")
import numpy as np

cfgs = {"2": ("NONE", "ANS0", 4 << 20), "3": ("BWT+MTFT+ZRLT", "ANS0", 8 << 20), "h": ("NONE", "HUFFMAN", 4 << 20),
        "1": ("NONE", "ANS1", 4 << 20), "4": ("BWT+SRT+ZRLT", "FPAQ", 4 << 20)}
which = sys.argv[1] if len(sys.argv) > 1 else "2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 0
t, e, bs = cfgs[which]
data, desc = corpus.load("silesia", limit or None)
n = len(data)
dev = torch.device("cuda", 0)
ctx = hipapi.Context(0)
d_in = torch.empty(n + 64, dtype=torch.uint8, device=dev)
d_in[:n].copy_(torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()))
p = ctx.params(t, e, bs)
cap = ctx.encode_bound(p, n)
d_enc = torch.zeros(cap, dtype=torch.uint8, device=dev)
d_dec = torch.empty(n + bs + 64, dtype=torch.uint8, device=dev)
first = None
bad = 0
t0 = time.time()
for i in range(reps):
    bits = ctx.encode_blocks(p, d_in.data_ptr(), n, d_enc.data_ptr(), cap)
    nb = (bits + 7) // 8
    cur = d_enc[:nb].clone()
    if first is None:
        first = cur
    elif cur.numel() != first.numel() or not torch.equal(cur, first):
        bad += 1; print("iteration", i, "ENCODE differs"); first = first
    d_dec.zero_()
    try:
        ob, eb, nblk = ctx.decode_blocks(p, d_enc.data_ptr(), bits, 0, d_dec.data_ptr(), n + bs)
        if ob != n or not torch.equal(d_dec[:n], d_in[:n]):
            bad += 1; print("iteration", i, "DECODE mismatch", ob)
    except Exception as ex:
        bad += 1; print("iteration", i, "DECODE exception", ex)
print("config", which, desc, "reps", reps, "bad", bad, "%.1f s" % (time.time() - t0))
sys.exit(1 if bad else 0)
