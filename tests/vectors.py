"""Deterministic test inputs shared by the golden generator and the tests. An input is a spec tuple."""
import numpy as np

import knzlib


def make(spec):
    kind = spec[0]
    c = knzlib.corpus()
    if kind == "text":
        return c.text(spec[1], spec[2])
    if kind == "mixed":
        return c.mixed(spec[1], spec[2])
    if kind == "mixedslice":          # mixed(total, seed)[a:b]
        return c.mixed(spec[1], spec[2])[spec[3]:spec[4]]
    if kind == "rand":
        return np.random.default_rng(spec[2]).integers(0, 256, spec[1], dtype=np.uint8).tobytes()
    if kind == "geom":                # skewed alphabet
        return bytes((np.random.default_rng(spec[2]).geometric(spec[3] / 100.0, spec[1]) % 256).astype(np.uint8))
    if kind == "const":
        return bytes([spec[2]]) * spec[1]
    if kind == "ramp":
        return bytes((np.arange(spec[1]) & 255).astype(np.uint8))
    if kind == "formula13":           # src/test/TestEntropyCodec.cpp:237-238
        i = np.arange(spec[1], dtype=np.int64)
        return bytes((((i * 13) ^ (i >> 3) ^ ((i & 15) << 4)) & 255).astype(np.uint8))
    if kind == "mul17":               # src/test/TestEntropyCodec.cpp:283-284
        return bytes(((np.arange(spec[1], dtype=np.int64) * 17) & 255).astype(np.uint8))
    if kind == "fill17":              # src/test/test_api.py fill_buffer: (i*17+3)&255
        return bytes(((np.arange(spec[1], dtype=np.int64) * 17 + 3) & 255).astype(np.uint8))
    if kind == "kat32":               # src/test/TestTransforms.cpp:889-899
        return bytes([0, 1, 2, 2, 2, 2, 7, 9, 9, 16, 16, 16, 1] + [3] * 19)
    if kind == "runs":
        return b"".join(bytes([i % 251]) * ((i * 37) % spec[1] + 1) for i in range(spec[2]))
    if kind == "twosym":
        return bytes((np.random.default_rng(spec[2]).integers(0, 2, spec[1]) * 7).astype(np.uint8))
    if kind == "ffmix":
        return bytes([0xFE, 0xFF, 0, 0, 0, 5, 0xFF, 0xFF] * spec[1])
    if kind == "fib":                 # Fibonacci frequencies: Huffman code lengths > 12 -> limitCodeLengths
        a, b, out = 1, 1, bytearray()
        for i in range(spec[1]):
            out += bytes([(i * 7 + 3) & 255]) * a
            a, b = b, a + b
        rng = np.random.default_rng(spec[2])
        arr = np.frombuffer(bytes(out), dtype=np.uint8).copy()
        rng.shuffle(arr)
        return arr.tobytes()
    if kind == "pow":                 # frequencies ~ ratio^i: long tails of rare symbols
        rng = np.random.default_rng(spec[3])
        w = np.array([spec[2] ** (-i / 10.0) for i in range(spec[4])])
        return bytes(rng.choice(spec[4], size=spec[1], p=w / w.sum()).astype(np.uint8))
    if kind == "repeats":
        return c.repeats(spec[1], spec[2])
    if kind == "tile":                # text(period, seed) repeated
        return c.tile(spec[1], spec[2], spec[3])
    if kind == "periodic":            # random unit of spec[3] bytes repeated
        return c.periodic(spec[1], spec[2], spec[3])
    if kind == "fibword":
        return c.fibword(spec[1])
    if kind == "dna":
        return c.dna(spec[1], spec[2])
    if kind == "records":             # rows of spec[3] bytes, spec[4] of them kept from the row before
        return c.records(spec[1], spec[2], spec[3], spec[4])
    if kind == "gradient":            # image rows of spec[3] samples
        return c.gradient(spec[1], spec[2], spec[3])
    if kind == "utf8":                # code points of one, two, three and four bytes, skewed: what the UTF transform takes
        rng = np.random.default_rng(spec[2])
        cps = [int(x) for x in rng.integers(0x80, 0x7FF, 60)] + [int(x) for x in rng.integers(0x800, 0xD7FF, 40)] + [0x1F600, 0x10348, 0x20AC] + list(range(97, 123)) * 2 + [32] * 12 + [10]
        w = np.array([1.0 / (1 + i % 37) for i in range(len(cps))])
        idx = rng.choice(len(cps), size=spec[1] // 2 + 16, p=w / w.sum())
        return "".join(chr(cps[int(i)]) for i in idx).encode("utf-8")[:spec[1]]
    if kind == "crlf":                # text with DOS line ends and capitalised words
        t = c.text(spec[1], spec[2]).replace(b".\n", b".\r\nThe ")
        return t[:spec[1]]
    if kind == "xml":                 # text inside tags, with entities
        t = c.text(spec[1], spec[2]).replace(b".\n", b"</p>\n<p class=\"x\">&amp; ")
        return (b"<doc>" + t)[:spec[1]]
    if kind == "bytes":
        return bytes.fromhex(spec[1])
    if kind == "str":
        return spec[1].encode()
    raise ValueError(kind)


STAGE_INPUTS = [
    ("kat32",), ("str", "mississippi"), ("str", "3.14159265358979323846264338327950288419716939937510"),
    ("str", "SIX.MIXED.PIXIES.SIFT.SIXTY.PIXIE.DUST.BOXES"),
    ("const", 33, 65), ("const", 70000, 0), ("ramp", 255), ("ramp", 256), ("ramp", 1000),
    ("formula13", 4096), ("mul17", 65536), ("fill17", 1024), ("text", 65536, 1), ("text", 16385, 4), ("text", 16387, 4),
    ("text", 4096, 3), ("mixedslice", 5 * 262144, 2, 3 * 262144, 3 * 262144 + 70000), ("mixed", 300000, 2),
    ("rand", 50000, 7), ("geom", 100001, 3, 30), ("twosym", 50000, 5), ("ffmix", 3000), ("runs", 9000, 60),
    ("str", "abcabcabcabcabcabcab"), ("ramp", 33), ("ramp", 15), ("ramp", 1),
    ("fib", 19, 1), ("fib", 20, 2), ("pow", 16384, 14, 3, 200), ("pow", 16000, 20, 4, 120), ("pow", 40000, 17, 5, 256),
]

STREAM_CASES = [
    # (input spec, transform, entropy, block size, checksum, headerless)
    (("text", 4194304, 1), "NONE", "HUFFMAN", 4 << 20, 0, 0),
    (("text", 4194304, 1), "NONE", "ANS0", 4 << 20, 0, 0),
    (("text", 4194304, 1), "BWT+MTFT+ZRLT", "ANS0", 1 << 20, 0, 0),
    (("text", 4194304, 1), "BWT+SRT+ZRLT", "FPAQ", 1 << 20, 0, 0),
    (("text", 4194304, 1), "RLT", "HUFFMAN", 1 << 20, 0, 0),
    (("mixed", 4194304, 2), "NONE", "HUFFMAN", 4 << 20, 0, 0),
    (("mixed", 4194304, 2), "NONE", "ANS0", 4 << 20, 0, 0),
    (("mixed", 4194304, 2), "BWT+MTFT+ZRLT", "ANS0", 1 << 20, 0, 0),
    (("mixed", 4194304, 2), "BWT+SRT+ZRLT", "FPAQ", 1 << 20, 0, 0),
    (("mixed", 4194304, 2), "RLT", "HUFFMAN", 1 << 20, 0, 0),
    (("mixed", 3 * (1 << 20) + 12345, 2), "NONE", "ANS0", 1 << 20, 0, 1),
    (("mixed", 3 * (1 << 20) + 12345, 2), "BWT+MTFT+ZRLT", "ANS0", 1 << 18, 32, 0),
    (("mixed", 1 << 20, 3), "ZRLT", "HUFFMAN", 16384, 64, 0),
    (("mixed", 1 << 20, 3), "SRT", "ANS1", 1 << 20, 0, 0),
    (("mixed", 1 << 20, 3), "RLT+ZRLT", "FPAQ", 1 << 19, 0, 0),
    (("mixed", 1 << 20, 3), "MTFT", "NONE", 1 << 20, 0, 0),
    (("rand", 300000, 9), "BWT+MTFT+ZRLT", "ANS0", 1 << 18, 0, 0),
    (("rand", 300000, 9), "ZRLT", "ANS0", 1 << 16, 0, 0),
    (("ramp", 0), "NONE", "ANS0", 1024, 0, 0),
    (("ramp", 1), "BWT+MTFT+ZRLT", "ANS0", 1024, 0, 0),
    (("ramp", 15), "BWT+MTFT+ZRLT", "ANS0", 1024, 0, 0),
    (("ramp", 16), "BWT+MTFT+ZRLT", "ANS0", 1024, 0, 0),
    (("ramp", 1025), "BWT+MTFT+ZRLT", "ANS0", 1024, 0, 0),
    (("const", 100000, 0), "BWT+MTFT+ZRLT", "ANS0", 65536, 0, 0),
    (("const", 100000, 0), "RLT", "NONE", 65536, 0, 0),
    (("text", 4194304, 1), "LZX", "ANS1", 1 << 20, 0, 0),
    (("mixed", 4194304, 2), "LZ", "HUFFMAN", 1 << 20, 0, 0),
    (("mixed", 1 << 20, 3), "RLT+LZX", "ANS0", 1 << 18, 32, 0),
    (("rand", 300000, 9), "LZX", "ANS0", 1 << 16, 0, 0),
    (("const", 100000, 0), "LZ", "NONE", 65536, 0, 0),
    (("mixed", 1 << 20, 3), "BWT+RANK+ZRLT", "ANS0", 1 << 18, 0, 0),
]

# BASELINE.json configurations at their own block sizes on 64 MiB of the stand-in corpora: the reference's .knz is
# pinned by md5 + length in tests/golden/golden_full.json (tests/golden/make_golden_full.py, run where oracle/_ref exists).
FULL_CASES = [
    # (config number, input spec, transform, entropy, block size)
    (1, ("mixed", 64 << 20, 2), "NONE", "HUFFMAN", 4 << 20),
    (2, ("mixed", 64 << 20, 2), "NONE", "ANS0", 4 << 20),
    (3, ("mixed", 64 << 20, 2), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    (4, ("text", 64 << 20, 1), "BWT+SRT+ZRLT", "FPAQ", 32 << 20),
    (5, ("mixed", 64 << 20, 2), "LZX", "ANS1", 16 << 20),
]

# Inputs with long common prefixes at the headline chain's own block size (8 MiB): what a prefix-doubling suffix sorter finds hard and
# what divsufsort (the reference) does not care about. Same fixture file, config "hard:<name>".
HARD_CASES = [
    ("hard:repeats", ("repeats", 16 << 20, 3), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:xx", ("tile", 16 << 20, 1, 4 << 20), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:period3", ("periodic", 8 << 20, 5, 3), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:period5", ("periodic", (8 << 20) - 3, 6, 5), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:period7", ("periodic", 8 << 20, 7, 7), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:period768", ("periodic", 8 << 20, 8, 768), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:period1500", ("periodic", 8 << 20, 9, 1500), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),      # medium groups of a period no power of two: k_bwt_f_probe
    ("hard:fibword", ("fibword", 8 << 20), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:dna", ("dna", 16 << 20, 4), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:const", ("const", 8 << 20, 65), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:repeats_srt", ("repeats", 32 << 20, 5), "BWT+SRT+ZRLT", "ANS0", 32 << 20),
    # shapes of silesia members that are no text: fixed-length records (osdb), an image plane (mr, x-ray)
    ("hard:records24", ("records", 8 << 20, 11, 24, 20), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:records100", ("records", 8 << 20, 12, 100, 93), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    ("hard:gradient", ("gradient", 8 << 20, 13, 1000), "BWT+MTFT+ZRLT", "ANS0", 8 << 20),
    # config 4's chain on four blocks of its own size: block ids 2 and 3 of a 32 MiB stream (slot model i % jobs, first_block_id)
    ("config4:4blocks", ("text", 128 << 20, 1), "BWT+SRT+ZRLT", "FPAQ", 32 << 20),
    # the same chain on SIX blocks of text with copied spans: block ids >= 4 of a 32 MiB stream, long common prefixes through SRT and FPAQ
    ("config4:6blocks_repeats", ("repeats", 6 * (32 << 20), 5), "BWT+SRT+ZRLT", "FPAQ", 32 << 20),
]

# Block sizes the reference accepts and the other cases never reach (io/CompressedOutputStream.cpp:69-82: up to 1 GiB; transform/BWT.cpp:32:
# MAX_BLOCK_SIZE 1 GiB): ONE block of 256 MiB through the headline chain, one block of 1 GiB through the entropy coders and through the
# suffix sorter alone -- 30-bit positions in the round-0 keys, the count + scatter passes instead of the one-sweep ones (RS_VAL_MASK),
# slot arithmetic near 2^30, the 2 GiB-per-call limit of the C ABI. Same fixture file, config "big:<name>" (VERDICT r5 item 5).
BIG_CASES = [
    ("big:bwt_chain_256m", ("mixed", 256 << 20, 3), "BWT+MTFT+ZRLT", "ANS0", 256 << 20),
    ("big:ans0_1g", ("mixed", 1 << 30, 4), "NONE", "ANS0", 1 << 30),
    ("big:huffman_1g", ("mixed", 1 << 30, 4), "NONE", "HUFFMAN", 1 << 30),
    # A BWT block of exactly 1 GiB is a stream the reference writes and cannot read: the block codec's header makes the stored length
    # 2^30 + 33, above the decoder's limit (io/CompressedInputStream.cpp:893-905, "Invalid compressed block length", ERR_READ_FILE).
    # The fixture records that (ref_decode_error); the round trip of the largest block both sides read is the case behind it.
    ("big:bwt_1g", ("mixed", 1 << 30, 4), "BWT", "NONE", 1 << 30),
    ("big:bwt_1g_less_64k", ("mixedslice", 1 << 30, 4, 0, (1 << 30) - 65536), "BWT", "NONE", (1 << 30) - 65536),
]

# The CLI's level presets that reach the device chain through host stages (TEXT + UTF): `kanzi -c -l N` of the reference, digests in
# tests/golden/levels.json (tests/golden/make_levels.py)
LEVEL_CASES = [
    # (level, input spec, extra CLI arguments)
    (5, ("text", 9 << 20, 1), []),
    (6, ("text", 17 << 20, 2), []),
    (5, ("mixed", 9 << 20, 2), []),
    (6, ("mixed", 9 << 20, 3), ["-b", "2m"]),
    (5, ("utf8", 5 << 20, 7), ["-b", "1m"]),
    (6, ("utf8", 3 << 20, 8), ["-b", "1m", "-x64"]),
    (5, ("crlf", 3 << 20, 4), ["-b", "1m", "-x"]),
    (6, ("xml", 3 << 20, 5), ["-b", "1m"]),
    (5, ("repeats", 5 << 20, 9), ["-b", "2m"]),
    (5, ("text", 1500, 3), []),
    (5, ("text", 700, 3), []),
    (5, ("rand", 300000, 5), ["-b", "64k"]),
]

# inputs of the host stages' own fixture (tests/golden/host_stages.json): text, DOS text, XML, UTF-8, words with the escape bytes in them,
# data the stages refuse (random, DNA, digits, the mixed stand-in), at sizes around the 1,024-byte minimum and well above it
HOST_STAGE_INPUTS = [
    ("text", 1024, 1), ("text", 1023, 1), ("text", 20000, 2), ("text", 300000, 3), ("text", 2000000, 4),
    ("crlf", 60000, 4), ("xml", 60000, 5), ("utf8", 60000, 7), ("utf8", 400000, 8), ("repeats", 200000, 9),
    ("mixed", 300000, 2), ("mixedslice", 5 * 262144, 2, 262144, 262144 + 100000), ("rand", 50000, 7), ("dna", 50000, 4),
    ("formula13", 4096), ("const", 5000, 32), ("str", "The quick brown fox \x0f jumps \x0e over The lazy dog. " * 60),
    ("str", "caf\u00e9 na\u00efve \u20ac100 \U0001F600 r\u00e9sum\u00e9 \u4e2d\u6587 " * 90),
]
