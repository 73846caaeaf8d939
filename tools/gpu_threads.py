# two threads, two compressor contexts sharing the device: results must equal the single-threaded ones
import sys, os, threading, importlib
sys.path.insert(0, "/root/repo/tests")
import knzlib, vectors
knzlib.load_pkg()
kz = importlib.import_module("kanzi_amd.kanzi")
O = knzlib.Oracle()
def job(idx, res):
    d = vectors.make(("mixed", 3000000 + 1000 * idx, 5 + idx))
    path = "/tmp/mt_%d.knz" % idx
    for rep in range(6):
        c = kz.Compressor(path, "BWT+MTFT+ZRLT" if idx % 2 else "NONE", "ANS0", 262144, jobs=2)
        for off in range(0, len(d), 262144): c.compress(d[off:off + 262144])
        c.close()
        rc, ref = O.compress(d, "BWT+MTFT+ZRLT" if idx % 2 else "NONE", "ANS0", 262144, jobs=2)
        ok = open(path, "rb").read() == ref
        dd = kz.Decompressor(path, 262144, jobs=2); got = bytearray()
        while True:
            part = dd.decompress(262144)
            if not part: break
            got += part
        dd.close()
        ok = ok and bytes(got) == d
        res[idx] = res.get(idx, True) and ok
res = {}
ts = [threading.Thread(target=job, args=(i, res)) for i in range(4)]
[t.start() for t in ts]; [t.join() for t in ts]
print("multithread", res)
sys.exit(0 if all(res.values()) and len(res) == 4 else 1)
