"""Randomised CPU fuzz of the BWT forward / inverse kernels on the emulator (tests/emu/*.cpp): random block sets with periodic
stretches, runs, text, noise and mixtures, random workgroup orders and strategy switches, checked against the oracle by the
harness itself (developer tool).   usage: emu_fuzz_bwt.py SEED SECONDS"""
import os, struct, subprocess, sys, tempfile, time, pathlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import knzlib, test_emu_kernels as T

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = np.random.default_rng(seed)
tmp = pathlib.Path(tempfile.mkdtemp())
fwd = T.build("bwt_fwd_emu", tmp)
inv = T.build("bwt_inv_emu", tmp)
c = knzlib.corpus()

def piece(n):
    k = int(rng.integers(0, 8))
    if k == 0: return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    if k == 1: return c.text(n, int(rng.integers(1, 999)))
    if k == 2: return bytes([int(rng.integers(0, 256))]) * n
    if k == 3:
        p = int(rng.integers(1, 700)); unit = rng.integers(0, 256, p, dtype=np.uint8).tobytes()
        return (unit * (n // p + 1))[:n]
    if k == 4: return bytes((np.arange(n) % int(rng.integers(2, 300))).astype(np.uint8))
    if k == 5: return rng.integers(0, int(rng.integers(2, 5)), n, dtype=np.uint8).tobytes()
    if k == 6:
        out = bytearray()
        while len(out) < n: out += bytes([int(rng.integers(0, 256))]) * int(rng.geometric(0.02))
        return bytes(out[:n])
    a = c.mixed(max(n, 4096) + 1000, int(rng.integers(1, 99)))
    o = int(rng.integers(0, 1000)); return a[o:o + n]

def block():
    n = int(rng.choice([3, 2, 5, 37, 300, 4096, 4097, 8192, 9000, 20000, 70000, 150000], p=[.03, .03, .04, .05, .1, .1, .1, .1, .15, .15, .1, .05]))
    parts, left = [], n
    while left > 0:
        m = int(min(left, max(1, rng.integers(1, max(2, n)))))
        parts.append(piece(m)); left -= m
    return b"".join(parts)[:n]

t0 = time.time(); cases = 0
while time.time() - t0 < budget:
    blocks = [block() for _ in range(int(rng.integers(1, 5)))]
    path = str(tmp / "case.bin")
    T.write_case(path, blocks)
    env = dict(os.environ, HIPEMU_ORDER=str(int(rng.integers(0, 3))))
    r = rng.random()
    if rng.random() < 0.1: env["KNZ_BWT_NO_RUN_ROUND"] = "1"
    if rng.random() < 0.1: env["KNZ_BWT_RUN_FALLBACK"] = "1"
    if rng.random() < 0.15: env["KNZ_BWT_NSYM"] = str(int(rng.integers(1, 6)))
    for exe in (fwd, inv):
        p = subprocess.run([exe, path], capture_output=True, text=True, timeout=1800, env=env)
        if p.returncode != 0:
            keep = "/tmp/emu_fuzz_fail_%d_%d.bin" % (seed, cases)
            os.replace(path, keep)
            print("FAIL", exe, keep, {k: v for k, v in env.items() if k.startswith(("KNZ", "HIPEMU"))}, (p.stdout + p.stderr)[-500:], flush=True)
            sys.exit(1)
    cases += 1
print("emu fuzz seed %d: %d cases ok in %.0f s" % (seed, cases, time.time() - t0))
