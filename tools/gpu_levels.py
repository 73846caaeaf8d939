"""Levels 5 and 6 of the CLI against the reference's files (developer tool, GPU box): kanzi_amd_cli -c -l N must write what `kanzi -c -l N`
writes (md5 in the fixture), and must read both. usage: gpu_levels.py"""
import hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests"))
import knzlib, vectors
recs = json.load(open(os.path.join(HERE, "..", "tests", "golden", "levels.json")))
cli = os.path.join(HERE, "..", "kanzi-cpp_amd", "kanzi_amd_cli")
bad = 0
with tempfile.TemporaryDirectory() as td:
    for r in recs:
        d = vectors.make(tuple(r["input"]))
        src = os.path.join(td, "in.bin"); open(src, "wb").write(d)
        out = os.path.join(td, "out.knz"); back = os.path.join(td, "back.bin")
        args = [cli, "-c", "-i", src, "-o", out, "-f", "-l", str(r["level"])] + r.get("extra", [])
        p = subprocess.run(args, capture_output=True, text=True)
        enc = open(out, "rb").read() if os.path.exists(out) else b""
        ok = p.returncode == 0 and len(enc) == r["out"]["len"] and hashlib.md5(enc).hexdigest() == r["out"]["md5"]
        p2 = subprocess.run([cli, "-d", "-i", out, "-o", back, "-f"], capture_output=True, text=True)
        rt = p2.returncode == 0 and open(back, "rb").read() == d
        print("level %d %-28s %s: %d -> %d bytes (reference %d)  identical to the reference's file: %s  round trip: %s %s" % (
            r["level"], r["input"][0] + str(r["input"][1]), " ".join(r.get("extra", [])), len(d), len(enc), r["out"]["len"], ok, rt, (p.stderr + p2.stderr)[-200:].replace("\n", " ") if not (ok and rt) else ""), flush=True)
        bad += 0 if (ok and rt) else 1
sys.exit(1 if bad else 0)
