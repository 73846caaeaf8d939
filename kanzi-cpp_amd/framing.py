"""Stream header of the kanzi bitstream v6 (host-side logic, no compute).

Mirrors CompressedOutputStream::writeHeader (io/CompressedOutputStream.cpp:277-342) and
CompressedInputStream::readHeader (io/CompressedInputStream.cpp:511-663): 32 b magic, 4 b version,
2 b checksum size, 5 b entropy id, 48 b transform ids, 28 b blockSize>>4, 2 b szMask,
16*szMask b original size, 15 b padding, 24 b checksum.
"""

MAGIC = 0x4B414E5A
VERSION = 6
_M32 = 0xFFFFFFFF


class HeaderError(ValueError):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


def _cksum(ck_size, etype, ttype, block_size, sz_mask, size):
    H = 0x1E35A7BD
    c = (H * ((0x01030507 * VERSION) & _M32)) & _M32
    c ^= (H * (~ck_size & _M32)) & _M32
    c ^= (H * (~etype & _M32)) & _M32
    nt = ~ttype & 0xFFFFFFFFFFFFFFFF
    c ^= (H * ((nt >> 32) & _M32)) & _M32
    c ^= (H * (nt & _M32)) & _M32
    c ^= (H * (~block_size & _M32)) & _M32
    if sz_mask:
        ns = ~size & 0xFFFFFFFFFFFFFFFF
        c ^= (H * ((ns >> 32) & _M32)) & _M32
        c ^= (H * (ns & _M32)) & _M32
    return ((c >> 23) ^ (c >> 3)) & 0xFFFFFF


def _cksum_old(ver, etype, ttype, block_size, sz_mask, size):
    """io/CompressedInputStream.cpp:622-645 for versions below 6: seeded with the bare version, no checksum size, 16 bits"""
    H = 0x1E35A7BD
    c = (H * ver) & _M32
    c ^= (H * (~etype & _M32)) & _M32
    nt = ~ttype & 0xFFFFFFFFFFFFFFFF
    c ^= (H * ((nt >> 32) & _M32)) & _M32
    c ^= (H * (nt & _M32)) & _M32
    c ^= (H * (~block_size & _M32)) & _M32
    if sz_mask:
        ns = ~size & 0xFFFFFFFFFFFFFFFF
        c ^= (H * ((ns >> 32) & _M32)) & _M32
        c ^= (H * (ns & _M32)) & _M32
    return ((c >> 23) ^ (c >> 3)) & 0xFFFF


def make_header(etype, ttype, block_size, checksum_bits=0, orig_size=0):
    """Returns (bytes, nbits)."""
    ck = {0: 0, 32: 1, 64: 2}[checksum_bits]
    sz_mask = 0 if (orig_size == 0 or orig_size >= (1 << 48)) else ((orig_size.bit_length() - 1) >> 4) + 1
    v, n = 0, 0

    def put(val, bits):
        nonlocal v, n
        v = (v << bits) | (val & ((1 << bits) - 1))
        n += bits

    put(MAGIC, 32); put(VERSION, 4); put(ck, 2); put(etype, 5); put(ttype, 48); put(block_size >> 4, 28)
    put(sz_mask, 2)
    if sz_mask:
        put(orig_size, 16 * sz_mask)
    put(0, 15)
    put(_cksum(ck, etype, ttype, block_size, sz_mask, orig_size), 24)
    nbytes = (n + 7) // 8
    return (v << (8 * nbytes - n)).to_bytes(nbytes, "big"), n


def parse_header(data):
    """Returns dict(etype, ttype, block_size, checksum_bits, orig_size, bits). Raises HeaderError(code)."""
    total = int.from_bytes(data[:32].ljust(32, b"\0"), "big")
    pos = 0

    def get(bits):
        nonlocal pos
        val = (total >> (256 - pos - bits)) & ((1 << bits) - 1)
        pos += bits
        return val

    if get(32) != MAGIC:
        raise HeaderError(15, "Invalid stream type")
    ver = get(4)
    if ver > VERSION:
        raise HeaderError(16, "Cannot read this version of the stream: %d" % ver)
    if ver >= 6:
        ck = get(2)
        if ck == 3:
            raise HeaderError(15, "Invalid bitstream, incorrect block checksum size")
    else:
        ck = get(1)                      # io/CompressedInputStream.cpp:555-558: one checksum flag before version 6
    etype = get(5)
    ttype = get(48)
    block_size = get(28) << 4
    if block_size < 1024 or block_size > (1 << 30):
        raise HeaderError(2, "Invalid bitstream, incorrect block size: %d" % block_size)
    sz_mask = get(2)
    size = get(16 * sz_mask) if sz_mask else 0
    if ver >= 6:
        get(15)
        c1, c2 = get(24), _cksum(ck, etype, ttype, block_size, sz_mask, size)
    else:
        c1, c2 = get(16), _cksum_old(ver, etype, ttype, block_size, sz_mask, size)
    if c1 != c2:
        raise HeaderError(19, "Invalid bitstream, header checksum mismatch")
    return dict(etype=etype, ttype=ttype, block_size=block_size, checksum_bits=32 * ck, orig_size=size, bits=pos, bs_version=ver)
