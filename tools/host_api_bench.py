"""End-to-end throughput of the drop-in C API (host buffers in, .knz file on tmpfs out, and back):
what a caller of libkanzi_amd.so sees, PCIe and host copies included (developer tool)."""
import sys, os, time, importlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as ge
ge.load_package()
kz = importlib.import_module("kanzi_amd.kanzi")
corpus = importlib.import_module("kanzi_amd.corpus")

cfg = {"2": ("NONE", "ANS0", 4 << 20), "3": ("BWT+MTFT+ZRLT", "ANS0", 8 << 20), "1": ("NONE", "HUFFMAN", 4 << 20)}
which = sys.argv[1] if len(sys.argv) > 1 else "2"
jobs_list = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [8]
t, e, bs = cfg[which]
data, desc = corpus.load("silesia", None)
n = len(data)
path = "/dev/shm/knz_host_bench.knz" if os.path.isdir("/dev/shm") else "/tmp/knz_host_bench.knz"
mv = memoryview(data)
for jobs in jobs_list:
    for rep in range(3):
        t0 = time.time()
        c = kz.Compressor(path, t, e, bs, jobs=jobs)
        for off in range(0, n, bs):
            c.compress(mv[off:off + bs])
        written = c.close()
        t1 = time.time()
        d = kz.Decompressor(path, bs, jobs=jobs)
        parts = []
        while True:
            part = d.decompress(bs)
            if not part:
                break
            parts.append(part)
        d.close()
        t2 = time.time()
        ok = b"".join(parts) == data
        print("config %s (%s/%s) jobs %d rep %d: compress %.0f MB/s, decompress %.0f MB/s, round trip %.0f MB/s, %d -> %d bytes, ok=%s" % (
            which, t, e, jobs, rep, n / (t1 - t0) / 1e6, n / (t2 - t1) / 1e6, n / (t2 - t0) / 1e6, n, written, ok), flush=True)
os.remove(path)
