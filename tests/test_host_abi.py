"""CPU-side checks of the product: the C-ABI library loads and exports every symbol declared in
include/knz_hip.h, and the host-side framing logic matches the oracle. No compute calls here."""
import importlib
import os
import re

import knzlib


def test_library_exports_declared_symbols():
    knzlib.load_pkg()
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    assert os.path.exists(hipapi.LIB_PATH), "run __graft_entry__.build() first"
    L = hipapi.lib()
    hdr = open(os.path.join(knzlib.ROOT, "include", "knz_hip.h")).read()
    declared = set(re.findall(r"KNZ_API\s+[\w\s\*]+?\b(knz_hip_\w+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(L, sym), sym
    assert declared == set(hipapi.SYMBOLS)


def test_host_library_exports_reference_c_api():
    # include/kanzi_api.h: the eight entry points of src/api/Compressor.hpp:80-116 / Decompressor.hpp:83-117
    knzlib.load_pkg()
    kz = importlib.import_module("kanzi_amd.kanzi")
    assert os.path.exists(kz.LIB_PATH), "run __graft_entry__.build() first"
    import ctypes
    L = ctypes.CDLL(kz.LIB_PATH)
    hdr = open(os.path.join(knzlib.ROOT, "include", "kanzi_api.h")).read()
    declared = set(re.findall(r"KANZI_API\s+[\w\s]+?\b(\w+)\s*\(", hdr))
    assert declared == set(kz.C_API_SYMBOLS)
    for sym in declared:
        assert hasattr(L, sym), sym
    assert L.getCompressorVersion() == 0x010000          # no GPU needed
    assert ctypes.sizeof(kz.cData) == 104 and ctypes.sizeof(kz.dData) == 120   # struct layouts of the reference headers


def test_header_matches_oracle(oracle):
    knzlib.load_pkg()
    fr = importlib.import_module("kanzi_amd.framing")
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    for t, e, bs, ck, sz in [("NONE", "ANS0", 4 << 20, 0, 0), ("BWT+MTFT+ZRLT", "ANS0", 8 << 20, 32, 211957760),
                             ("BWT+SRT+ZRLT", "FPAQ", 32 << 20, 64, 10 ** 9), ("RLT", "HUFFMAN", 1024, 0, 5),
                             ("NONE", "NONE", 1 << 30, 0, (1 << 48) - 1), ("NONE", "NONE", 1024, 0, 1 << 48)]:
        rc, o = oracle.compress(b"", t, e, bs, checksum=ck, orig_size=sz)
        assert rc == 0
        h, n = fr.make_header(hipapi.ENTROPY_IDS[e], hipapi.transform_type(t), bs, ck, sz)
        assert n % 8 == 0 and o[:n // 8] == h
        p = fr.parse_header(o)
        assert (p["etype"], p["ttype"], p["block_size"], p["checksum_bits"], p["orig_size"], p["bits"]) == (
            hipapi.ENTROPY_IDS[e], hipapi.transform_type(t), bs, ck, sz if sz < (1 << 48) else 0, n)


def test_transform_type_parse(oracle):
    knzlib.load_pkg()
    hipapi = importlib.import_module("kanzi_amd.hipapi")
    for names in ["NONE", "BWT", "BWT+MTFT+ZRLT", "BWT+SRT+ZRLT", "RLT+ZRLT", "NONE+ZRLT"]:
        assert hipapi.transform_type(names) == oracle.L.knzo_transform_type(names.encode())


def test_corpus_generators_are_pinned():
    import hashlib
    c = knzlib.corpus()
    assert hashlib.md5(c.text(4194304, 1)).hexdigest() == "533763267af795f681817771bd17d0cc"
    assert hashlib.md5(c.mixed(4194304, 2)).hexdigest() == "4f3716bf4e8db9d931141d3c144dfc8c"
    assert c.mixed(600000, 2) == c.mixed(4194304, 2)[:600000]


def test_radix_tile_size_is_one_value_across_the_library():
    """The radix-sort kernels (csrc/prims.hpp) are templates whose symbol names do not carry the tile size; two object files built with
    different sizes once shared one kernel symbol and faulted on the device (DESIGN_HISTORY.md section 7, round 4). The size is a header constant now and
    every translation unit that includes the header emits rs_tile_tag<keys>(): exactly one such name may exist in the library, every
    object file with radix kernels must carry it, and every k_rs_* instantiation must have a single definition in the .so."""
    import glob
    import subprocess
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm" if os.path.exists("/opt/rocm/lib/llvm/bin/llvm-nm") else "nm"
    so = os.path.join(knzlib.PKG, "libknz_hip.so")
    assert os.path.exists(so), "run __graft_entry__.build() first"

    def syms(path):
        out = subprocess.run([nm, "-C", path], capture_output=True, text=True, check=True).stdout
        return [ln.split(None, 2) for ln in out.splitlines() if len(ln.split(None, 2)) == 3]

    tags = {name for _, _, name in syms(so) if "rs_tile_tag<" in name}
    assert len(tags) == 1, tags
    assert re.search(r"rs_tile_tag<8192u?>", next(iter(tags))), tags
    defs = {}
    for _, kind, name in syms(so):
        if "prims::k_rs_" in name and "__device_stub__" not in name and kind in "TtWwVv":
            defs[name] = defs.get(name, 0) + 1
    assert defs and all(v == 1 for v in defs.values()), {k: v for k, v in defs.items() if v != 1}
    for obj in glob.glob(os.path.join(knzlib.PKG, "csrc", "*.o")):
        names = [name for _, _, name in syms(obj)]
        if any("prims::k_rs_" in n for n in names):
            t = {n for n in names if "rs_tile_tag<" in n}
            assert t == tags, (obj, t)
