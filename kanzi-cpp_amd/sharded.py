"""Multi-GPU sharding of one stream: blocks are independent, so rank r encodes a
contiguous range of block indices as one bit run on its own GPU; there is no collective in the
data path. The only exchange is the gather of the finished runs (variable-length byte strings) to
the writer rank, which concatenates them at bit granularity on the host -- exactly the ordered
append CompressedOutputStream performs (io/CompressedOutputStream.cpp:835-868), just per run
instead of per block.

`encode_run(data, first_block, with_header, finish) -> (bytes, nbits)` is the per-rank encoder:
the product passes DeviceRunEncoder (GPU, knz_hip_encode_blocks); tests inject a CPU stand-in.
"""
import importlib


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
    """runs: [(bytes, nbits)] in stream order -> (bytes, nbits) MSB-first."""
    acc, total = 0, 0
    for data, nbits in runs:
        if nbits == 0:
            continue
        v = int.from_bytes(data[:(nbits + 7) // 8], "big") >> ((-nbits) % 8)
        acc = (acc << nbits) | v
        total += nbits
    nbytes = (total + 7) // 8
    return (acc << ((-total) % 8)).to_bytes(nbytes, "big") if nbytes else b"", total


class DeviceRunEncoder:
    """Encodes a run of blocks on this rank's GPU (no CPU fallback)."""

    def __init__(self, device, transform, entropy, block_size, jobs=1, orig_size=0, headerless=False):
        hipapi = importlib.import_module("kanzi_amd.hipapi")
        framing = importlib.import_module("kanzi_amd.framing")
        self.ctx = hipapi.Context(device)
        self.p = self.ctx.params(transform, entropy, block_size, 0, jobs)
        self.header = (b"", 0) if headerless else framing.make_header(self.p.entropy_type, self.p.transform_type, block_size, 0, orig_size)

    def __call__(self, data, first_block, with_header, finish):
        ctx = self.ctx
        hdr, hb = self.header if with_header else (b"", 0)
        cap = ctx.encode_bound(self.p, len(data)) + 64
        d_in, d_out = ctx.malloc(len(data) + 64), ctx.malloc(cap)
        try:
            if data:
                ctx.h2d(d_in, data)
            bits = ctx.encode_blocks(self.p, d_in, len(data), d_out, cap, prologue=hdr, prologue_bits=hb, first_block=first_block,
                                     finish=1 if finish else 0)
            return ctx.d2h(d_out, (bits + 7) // 8), bits
        finally:
            ctx.free(d_in)
            ctx.free(d_out)


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
