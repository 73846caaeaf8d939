"""End-to-end throughput of the drop-in C API under the host layer's knobs (developer tool, GPU box): bench.py's `end_to_end`
measurement for one configuration under every combination of KNZ_LANES x KNZ_BATCH_BLOCKS (and whatever other KNZ_* variables the
command line sets), with KNZ_HOST_TIMING=1 so that the stream classes print where their wall time went.
usage: python tools/host_e2e_sweep.py <config> <lanes,lanes,...> <batch,batch,...> [ENV=VALUE ...]"""
import importlib
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("KNZ_HOST_TIMING", "1")
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

ge.load_package()
corpus = importlib.import_module("kanzi_amd.corpus")
cfg_id = int(sys.argv[1]) if len(sys.argv) > 1 else 3
lanes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [4]
batches = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
for kv in sys.argv[4:]:
    k, v = kv.split("=", 1)
    os.environ[k] = v
cfg = bench.CONFIGS[cfg_id]
data, desc = corpus.load(cfg["corpus"], None)
for ln in lanes:
    for b in batches:
        os.environ["KNZ_LANES"] = str(ln)
        if b > 0:
            os.environ["KNZ_BATCH_BLOCKS"] = str(b)
        else:
            os.environ.pop("KNZ_BATCH_BLOCKS", None)
        sys.stderr.write("---- config %d lanes %d batch %s\n" % (cfg_id, ln, b or "default"))
        sys.stderr.flush()
        r = bench.end_to_end(data, cfg)
        print(json.dumps({"config": cfg_id, "lanes": ln, "batch_blocks": b, "value": r["value"], "compress": r["compress_MBps"], "decompress": r["decompress_MBps"]}), flush=True)
