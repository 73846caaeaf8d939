"""Multi-GPU sharding of one stream: blocks are independent, so rank r encodes a
contiguous range of block indices as one bit run on its own GPU; there is no collective in the
data path. The only exchange is the gather of the finished runs (variable-length byte strings) to
the writer rank, which concatenates them at bit granularity on the host -- exactly the ordered
append CompressedOutputStream performs (io/CompressedOutputStream.cpp:835-868), just per run
instead of per block.

`encode_run(data, first_block, with_header, finish) -> (bytes, nbits)` is the per-rank encoder:
the product passes DeviceRunEncoder (GPU, knz_hip_encode_blocks); tests inject a CPU stand-in.
Decode (`decompress_sharded`): every rank walks the block length prefixes on the host
(io/CompressedInputStream.cpp:823-856), decodes its own contiguous range of blocks with
`decode_run` (DeviceRunDecoder: knz_hip_decode_blocks) and rank 0 places the ranges in order.
"""
import importlib

import numpy as np


def block_ranges(n_bytes, block_size, world):
    """Contiguous, balanced ranges of block indices: [(first_block, n_blocks)] per rank."""
    nblocks = (n_bytes + block_size - 1) // block_size
    base, extra = divmod(nblocks, world)
    out, first = [], 0
    for r in range(world):
        cnt = base + (1 if r < extra else 0)
        out.append((first, cnt))
        first += cnt
    return out


def concat_bit_runs(runs):
    """runs: [(bytes, nbits)] in stream order -> (bytes, nbits) MSB-first. A run that does not start on a byte boundary is moved
    by 1..7 bits with two numpy shifts over its bytes (what knz_hip_shift_bits does on the device for the stream classes)."""
    total = sum(nbits for _, nbits in runs)
    out = np.zeros((total + 7) // 8 + 1, dtype=np.uint8)
    pos = 0
    for data, nbits in runs:
        if nbits == 0:
            continue
        nb = (nbits + 7) // 8
        src = np.frombuffer(data, dtype=np.uint8, count=nb)
        if nbits & 7:                                   # bits behind the run's end do not belong to it
            src = src.copy()
            src[-1] &= (0xFF << (8 - (nbits & 7))) & 0xFF
        r, at = pos & 7, pos >> 3
        if r == 0:
            out[at:at + nb] |= src
        else:
            out[at:at + nb] |= src >> r
            out[at + 1:at + 1 + nb] |= (src << (8 - r)).astype(np.uint8)
        pos += nbits
    return out[:(total + 7) // 8].tobytes(), total


class _DeviceBuffers:
    """Device staging buffers of one rank, grown on demand and kept across calls."""

    def __init__(self, ctx):
        self.ctx, self.ptr, self.cap = ctx, {}, {}

    def get(self, name, nbytes):
        if self.cap.get(name, 0) < nbytes:
            if name in self.ptr:
                self.ctx.free(self.ptr[name])
            want = nbytes + (nbytes >> 2) + 4096
            self.ptr[name], self.cap[name] = self.ctx.malloc(want), want
        return self.ptr[name]


class DeviceRunEncoder:
    """Encodes a run of blocks on this rank's GPU (no CPU fallback)."""

    def __init__(self, device, transform, entropy, block_size, jobs=1, orig_size=0, headerless=False):
        hipapi = importlib.import_module("kanzi_amd.hipapi")
        framing = importlib.import_module("kanzi_amd.framing")
        self.ctx = hipapi.Context(device)
        self.p = self.ctx.params(transform, entropy, block_size, 0, jobs)
        self.header = (b"", 0) if headerless else framing.make_header(self.p.entropy_type, self.p.transform_type, block_size, 0, orig_size)
        self.bufs = _DeviceBuffers(self.ctx)

    def __call__(self, data, first_block, with_header, finish):
        ctx = self.ctx
        hdr, hb = self.header if with_header else (b"", 0)
        cap = ctx.encode_bound(self.p, len(data)) + 64
        d_in, d_out = self.bufs.get("in", len(data) + 64), self.bufs.get("out", cap)
        if data:
            ctx.h2d(d_in, data)
        bits = ctx.encode_blocks(self.p, d_in, len(data), d_out, cap, prologue=hdr, prologue_bits=hb, first_block=first_block,
                                 finish=1 if finish else 0)
        return ctx.d2h(d_out, (bits + 7) // 8), bits


class DeviceRunDecoder:
    """Decodes a run of blocks on this rank's GPU: `chunk` holds the run, the first block's 5-bit length
    prefix sits at bit `start_bit` of it (no CPU fallback)."""

    def __init__(self, device, jobs=1):
        hipapi = importlib.import_module("kanzi_amd.hipapi")
        self.ctx = hipapi.Context(device)
        self.jobs = jobs
        self.bufs = _DeviceBuffers(self.ctx)

    def __call__(self, chunk, start_bit, end_bit, n_blocks, hdr):
        ctx = self.ctx
        p = ctx.params(hdr["ttype"], hdr["etype"], hdr["block_size"], hdr["checksum_bits"], self.jobs, hdr.get("bs_version", 6) or 1)      # a parsed version 0 is an old layout (the C ABI keeps 0 for "unset = current")
        out_cap = n_blocks * hdr["block_size"] + 64
        d_in, d_out = self.bufs.get("in", len(chunk) + 64), self.bufs.get("out", out_cap)
        ctx.h2d(d_in, chunk)
        nbytes, _, done = ctx.decode_blocks(p, d_in, end_bit, start_bit, d_out, out_cap, max_blocks=n_blocks)
        if done != n_blocks:
            raise RuntimeError("run of %d blocks: only %d decoded" % (n_blocks, done))
        return ctx.d2h(d_out, nbytes)


def _bits_at(buf, pos, n):
    """n <= 56 bits at bit position pos of an MSB-first byte string (zero past the end)."""
    first = pos >> 3
    word = int.from_bytes(bytes(buf[first:first + 8]).ljust(8, b"\0"), "big")
    return (word >> (64 - (pos & 7) - n)) & ((1 << n) - 1)


def walk_blocks(stream, start_bit):
    """The host-side walk over the block length prefixes (io/CompressedInputStream.cpp:823-856): 5 bits lw-3, lw
    bits length, then that many payload bits; a zero length ends the stream. Returns (positions, end) where
    positions[i] is the bit position of block i's prefix and `end` the position of the end marker's prefix (or the
    position where the stream stops when there is none)."""
    total, pos, out = 8 * len(stream), start_bit, []
    while pos + 8 <= total:
        lw = 3 + _bits_at(stream, pos, 5)
        if pos + 5 + lw > total:
            raise ValueError("Unexpected end of stream")
        length = _bits_at(stream, pos + 5, lw)
        if length == 0:
            break
        if pos + 5 + lw + length > total:
            raise ValueError("Unexpected end of stream")
        out.append(pos)
        pos += 5 + lw + length
    return out, pos


def decompress_sharded(stream, rank, world, decode_run, gather):
    """Decode side of section 8(e): every rank walks the length prefixes of `stream` (cheap, host only), takes a
    contiguous range of blocks, decodes it with `decode_run(chunk, start_bit, end_bit, n_blocks, hdr) -> bytes`
    (DeviceRunDecoder on a GPU) and the writer rank (0) places the ranges in order. Returns the plain bytes on
    rank 0, None elsewhere."""
    framing = importlib.import_module("kanzi_amd.framing")
    hdr = framing.parse_header(stream)
    positions, end = walk_blocks(stream, hdr["bits"])
    nblocks = len(positions)
    base, extra = divmod(nblocks, world)
    first = rank * base + min(rank, extra)
    cnt = base + (1 if rank < extra else 0)
    part = b""
    if cnt:
        lo_bit = positions[first]
        hi_bit = positions[first + cnt] if first + cnt < nblocks else end
        lo_byte = (lo_bit >> 3) & ~15                 # device buffers are read as aligned words
        hi_byte = (hi_bit + 7) >> 3
        part = decode_run(stream[lo_byte:hi_byte], lo_bit - 8 * lo_byte, hi_bit - 8 * lo_byte, cnt, hdr)
    parts = gather(part)
    if parts is None:
        return None
    return b"".join(parts)


def compress_sharded(data, block_size, rank, world, encode_run, gather):
    """Every rank passes the same `data` view (or at least its own slice) and gets None, except the
    writer (rank 0) which gets the complete stream. `gather(obj)` returns the list of all ranks'
    objects on rank 0 (e.g. torch.distributed.gather_object) and None elsewhere."""
    first, cnt = block_ranges(len(data), block_size, world)[rank]
    lo = first * block_size
    hi = min(len(data), (first + cnt) * block_size)
    last_rank_with_blocks = max([r for r, (f, c) in enumerate(block_ranges(len(data), block_size, world)) if c > 0] + [0])
    run = encode_run(data[lo:hi] if cnt else b"", first, rank == 0, rank == last_rank_with_blocks)
    runs = gather(run)
    if runs is None:
        return None
    runs = [r for i, r in enumerate(runs) if i <= last_rank_with_blocks]
    return concat_bit_runs(runs)[0]
