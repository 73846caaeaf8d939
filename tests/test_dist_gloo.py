"""N>1 path on CPU: world_size-2 gloo run of the block sharding + host-side bit concatenation, and of the decode
sharding (host prefix walk -> per-rank block ranges -> ordered placement)
(kanzi-cpp_amd/sharded.py). The per-rank GPU encoder is replaced by a CPU stand-in built on the
oracle (test infrastructure); what is under test is the partition, the gather and the ordered
bit-granular append, which must reproduce the single-process stream bit for bit."""
import importlib
import os
import subprocess
import sys

import knzlib
import vectors

WORKER = r"""
import importlib, os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import knzlib, vectors
import torch.distributed as dist
knzlib.load_pkg()
sharded = importlib.import_module("kanzi_amd.sharded")
framing = importlib.import_module("kanzi_amd.framing")
hipapi = importlib.import_module("kanzi_amd.hipapi")
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=int(sys.argv[4]))
rank, world = dist.get_rank(), dist.get_world_size()
O = knzlib.Oracle()
ok = True
for spec, t, e, bs, jobs in [(("mixed", 700001, 11), "BWT+MTFT+ZRLT", "ANS0", 65536, 1), (("text", 300000, 3), "NONE", "HUFFMAN", 16384, 2),
                             (("mixed", 100000, 5), "BWT+SRT+ZRLT", "FPAQ", 1 << 20, 1), (("ramp", 0), "NONE", "ANS0", 1024, 1)]:
    data = vectors.make(spec)
    hdr = framing.make_header(hipapi.ENTROPY_IDS[e], hipapi.transform_type(t), bs, 0, len(data))
    def encode_run(chunk, first_block, with_header, finish):
        rc, out, bits = O.compress_run(chunk, t, e, bs, first_block, finish, jobs=jobs)
        assert rc == 0
        if with_header:
            return sharded.concat_bit_runs([hdr, (out, bits)])
        return out, bits
    def gather(obj):
        lst = [None] * world if rank == 0 else None
        dist.gather_object(obj, lst, dst=0)
        return lst
    res = sharded.compress_sharded(data, bs, rank, world, encode_run, gather)
    if rank == 0:
        rc, ref = O.compress(data, t, e, bs, orig_size=len(data), jobs=jobs)
        if res != ref:
            ok = False
            print("MISMATCH", spec, t, e, len(res), len(ref))
    # decode side: every rank walks the prefixes of the reference stream and decodes its own range of blocks
    rc, ref = O.compress(data, t, e, bs, orig_size=len(data), jobs=jobs)
    def decode_run(chunk, start_bit, end_bit, n_blocks, h):
        # CPU stand-in for the GPU decoder: header + this run's bits + end marker is a complete stream for the oracle
        v = int.from_bytes(chunk, "big") >> (8 * len(chunk) - end_bit)
        nb = end_bit - start_bit
        v &= (1 << nb) - 1
        run = (v << ((-nb) % 8)).to_bytes((nb + 7) // 8, "big")
        stream, _ = sharded.concat_bit_runs([framing.make_header(h["etype"], h["ttype"], h["block_size"], h["checksum_bits"], 0), (run, nb), (b"\0", 8)])
        rc2, out = O.decompress(stream, n_blocks * h["block_size"])
        assert rc2 == 0, rc2
        return out
    back = sharded.decompress_sharded(ref, rank, world, decode_run, gather)
    if rank == 0 and back != data:
        ok = False
        print("DECODE MISMATCH", spec, t, e, len(back), len(data))
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
"""


def test_block_ranges_and_bit_concat():
    knzlib.load_pkg()
    sh = importlib.import_module("kanzi_amd.sharded")
    assert sh.block_ranges(10 * 1024 + 1, 1024, 4) == [(0, 3), (3, 3), (6, 3), (9, 2)]
    assert sh.block_ranges(0, 1024, 2) == [(0, 0), (0, 0)]
    assert sh.concat_bit_runs([(b"\xA0", 3), (b"\xFF", 8), (b"\x80", 1)]) == (bytes([0b10111111, 0b11110000]), 12)
    # prefix walk: two blocks (lw = 3: payload lengths 5 and 3 bits), then the end marker
    bits = "00000" + "101" + "11111" + "00000" + "011" + "101" + "00000" + "000"
    stream = int(bits.ljust(40, "0"), 2).to_bytes(5, "big")
    assert sh.walk_blocks(stream, 0) == ([0, 13], 24)


def test_two_rank_gloo_sharded_stream(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = str(29500 + (os.getpid() % 500))
    procs = [subprocess.Popen([sys.executable, str(script), knzlib.ROOT, port, str(r), "2"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
