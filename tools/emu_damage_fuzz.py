"""Damaged input against the emulated decoder / inverse kernels, built with AddressSanitizer (developer tool; the suite runs a short
form: tests/test_emu_kernels.py::test_decoders_survive_damaged_input_emulated).   usage: emu_damage_fuzz.py FIRST_SEED SECONDS"""
import os, subprocess, sys, tempfile, time, pathlib
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np
import knzlib, test_emu_kernels as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
tmp = pathlib.Path(tempfile.mkdtemp())
runs = [("huff_emu", ["6", "5"]), ("ans0_emu", ["6"]), ("ans1_emu", ["6"]), ("fpaq_emu", ["6"]), ("bwt_inv_emu", ["6", "5"]), ("lz_emu", ["6", "5"]),
        ("lzx_emu", ["6"]), ("srt_emu", ["6"]), ("zrlt_emu", ["6"]), ("mtft_emu", ["6"]), ("rlt_emu", ["6"]), ("rank_emu", ["6"])]
c = knzlib.corpus()
with ThreadPoolExecutor(max_workers=6) as pool:
    exes = dict(zip([n for n, _ in runs], pool.map(lambda n: T.build(n, tmp, extra=["-fsanitize=address", "-g", "-fno-omit-frame-pointer"]), [n for n, _ in runs])))
    t0 = time.time(); seed = first; cases = 0; bad = 0
    while time.time() - t0 < budget:
        rng = np.random.default_rng(seed)
        blocks = [c.text(int(rng.integers(100, 30000)), seed), rng.integers(0, 256, int(rng.integers(50, 9000)), dtype=np.uint8).tobytes(),
                  c.mixed(300000, seed % 90 + 1)[250000:250000 + int(rng.integers(300, 20000))], rng.integers(0, int(rng.integers(2, 9)), 5000, dtype=np.uint8).tobytes(),
                  bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 5000)), c.text(4099, seed + 1)]
        path = str(tmp / ("d%d.bin" % seed))
        T.write_case(path, blocks)

        def one(job):
            name, ver = job
            env = dict(os.environ, EMU_CORRUPT=str(seed), ASAN_OPTIONS="detect_leaks=0:detect_stack_use_after_return=0")
            try:
                r = subprocess.run([exes[name], path] + ([ver] if ver != "6" else []), capture_output=True, text=True, timeout=900, env=env)
                return job, r.returncode, r.stderr
            except subprocess.TimeoutExpired:
                return job, -999, "timeout"
        for job, rc, err in pool.map(one, [(n, v) for n, vs in runs for v in vs]):
            cases += 1
            if rc != 0 or "AddressSanitizer" in err:
                bad += 1
                keep = "/tmp/emu_damage_fail_%d_%s_%s.bin" % (seed, job[0], job[1])
                os.replace(path, keep) if os.path.exists(path) else None
                print("FAIL", job, "seed", seed, "rc", rc, keep, err[-1500:], flush=True)
        if os.path.exists(path): os.remove(path)
        seed += 1
print("emu damage fuzz seeds %d..%d: %d runs, %d bad in %.0f s" % (first, seed - 1, cases, bad, time.time() - t0))
