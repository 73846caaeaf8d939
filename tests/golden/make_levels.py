#!/usr/bin/env python3
"""Generates tests/golden/levels.json: length + md5 of what the UNMODIFIED reference CLI (oracle/_ref/kanzi) writes for the level presets
that run TEXT / UTF on the host in front of the device chain (vectors.LEVEL_CASES), -j 1.

    make -C oracle ref && python tests/golden/make_levels.py

Data only (input specs + digests of the reference's output)."""
import hashlib, json, os, subprocess, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import knzlib, vectors  # noqa: E402

def main():
    assert knzlib.ensure_ref() is not None
    out = []
    with tempfile.TemporaryDirectory() as td:
        for level, spec, extra in vectors.LEVEL_CASES:
            d = vectors.make(spec)
            src, dst, back = os.path.join(td, "in.bin"), os.path.join(td, "out.knz"), os.path.join(td, "back.bin")
            open(src, "wb").write(d)
            subprocess.check_call([knzlib.REF_BIN, "-c", "-i", src, "-o", dst, "-f", "-j", "1", "-l", str(level)] + extra, stdout=subprocess.DEVNULL)
            subprocess.check_call([knzlib.REF_BIN, "-d", "-i", dst, "-o", back, "-f", "-j", "1"], stdout=subprocess.DEVNULL)
            assert open(back, "rb").read() == d
            o = open(dst, "rb").read()
            out.append({"level": level, "input": list(spec), "extra": extra, "input_md5": hashlib.md5(d).hexdigest(), "out": {"len": len(o), "md5": hashlib.md5(o).hexdigest()}})
            print(out[-1], flush=True)
    json.dump(out, open(os.path.join(HERE, "levels.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
