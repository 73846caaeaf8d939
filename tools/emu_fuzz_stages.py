"""Randomised CPU fuzz of the emulated stage kernels (tests/emu/*_emu.cpp): MTFT, ZRLT, ANS0, ANS1 and Huffman in both directions, SRT,
RLT -- random blocks made of text, noise, runs, periodic and sparse pieces, checked against the oracle by the harnesses themselves
(developer tool).   usage: emu_fuzz_stages.py SEED SECONDS [harness,harness,...]   e.g. lz_emu,lzx_emu for the LZ kernels (both block
layouts: the harness is run with and without the old-layout argument)"""
import os, subprocess, sys, tempfile, time, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import knzlib, test_emu_kernels as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
tmp = pathlib.Path(tempfile.mkdtemp())
names = ["mtft_emu", "zrlt_emu", "ans0_emu", "ans0_enc_emu", "ans1_emu", "ans1_enc_emu", "huff_emu", "huff_enc_emu", "srt_emu", "rlt_emu"]
if len(sys.argv) > 3: names = sys.argv[3].split(",")
exes = {n: T.build(n, tmp) for n in names}
c = knzlib.corpus()

def piece(n):
    k = int(rng.integers(0, 8))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 1: return c.text(n, int(rng.integers(1, 999)))
    if k == 2: return bytes([int(rng.integers(0, 256))]) * n
    if k == 3:
        p = int(rng.integers(1, 700)); unit = rng.integers(0, 256, p, dtype=np.uint8).tobytes()
        return (unit * (n // p + 1))[:n]
    if k == 4:
        a = np.zeros(n, dtype=np.uint8); m = max(1, n // int(rng.integers(4, 200)))
        a[rng.integers(0, n, m)] = rng.integers(1, 256, m, dtype=np.uint8); return a.tobytes()
    if k == 5: return rng.integers(0, int(rng.integers(2, 20)), n, dtype=np.uint8).tobytes()
    if k == 6: return bytes([0xFF, 0xFE, 0x00][int(rng.integers(0, 3))] for _ in range(min(n, 50))) * (n // 50 + 1)
    a = c.mixed(max(n, 4096) + 1000, int(rng.integers(1, 99))); o = int(rng.integers(0, 1000)); return a[o:o + n]

def block():
    n = int(rng.choice([1, 2, 31, 32, 33, 300, 4095, 4096, 4097, 16383, 16384, 16385, 40000, 70000]))
    parts, left = [], n
    while left > 0:
        m = int(min(left, max(1, rng.integers(1, max(2, n))))); parts.append(piece(m)[:m]); left -= m
    return b"".join(parts)[:n]

t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    blocks = [block() for _ in range(int(rng.integers(1, 6)))]
    path = str(tmp / "case.bin")
    T.write_case(path, blocks)
    for n in names:
        extra = ["5"] if n in ("lz_emu", "lzx_emu") and int(rng.integers(0, 3)) == 0 else []
        p = subprocess.run([exes[n], path] + extra, capture_output=True, text=True, timeout=1800, env=dict(os.environ, HIPEMU_ORDER=str(int(rng.integers(0, 3)))))
        if p.returncode != 0:
            keep = "/tmp/emu_fuzz_stage_fail_%d_%d.bin" % (seed, cases)
            os.replace(path, keep)
            print("FAIL", n, keep, (p.stdout + p.stderr)[-400:], flush=True)
            sys.exit(1)
    cases += 1
print("emu stage fuzz seed %d: %d cases ok in %.0f s" % (seed, cases, time.time() - t0))
