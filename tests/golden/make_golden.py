#!/usr/bin/env python3
"""Generates tests/golden/golden.json from the UNMODIFIED reference compiled into oracle/_ref/
(run in the build container, where /root/reference exists):

    make -C oracle ref && python tests/golden/make_golden.py

The fixture holds data only: input specs (tests/vectors.py) and the reference's outputs (hex for
small results, md5 + length otherwise). Nothing of the reference's sources is stored.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import knzlib  # noqa: E402
import vectors  # noqa: E402


def pack(b):
    if len(b) <= 512:
        return {"len": len(b), "hex": b.hex()}
    return {"len": len(b), "md5": hashlib.md5(b).hexdigest()}


def main():
    R = knzlib.Ref()
    out = {"stages": [], "streams": []}
    for spec in vectors.STAGE_INPUTS:
        d = vectors.make(spec)
        for e in ["NONE", "HUFFMAN", "ANS0", "ANS1", "FPAQ"]:
            enc, bits = R.entropy_encode(e, d)
            out["stages"].append({"kind": "entropy", "name": e, "input": list(spec), "bits": bits, "out": pack(enc)})
        for t in ["BWT", "MTFT", "ZRLT", "SRT", "RLT", "LZ", "LZX", "RANK"]:
            for ent in (["", "ANS0", "FPAQ"] if t == "RLT" else [""]):
                cap = len(d) if t == "ZRLT" else len(d) + 2048
                if t in ("LZ", "LZX"):
                    cap = len(d) + len(d) // 64 + 64          # >= getMaxEncodedLength, or the codec refuses
                ok, o, sk = R.forward(t, d, cap, ent or None)
                rec = {"kind": "transform", "name": t, "entropy": ent, "input": list(spec), "cap": cap, "ok": int(ok == 1)}
                if ok == 1:
                    rec["out"] = pack(o)
                out["stages"].append(rec)
    for spec, t, e, bs, ck, hl in vectors.STREAM_CASES:
        d = vectors.make(spec)
        rc, o = R.compress(d, t, e, bs, jobs=1, checksum=ck, orig_size=0 if hl else len(d), headerless=hl)
        assert rc == 0, (spec, t, e, rc)
        out["streams"].append({"input": list(spec), "transform": t, "entropy": e, "block": bs, "checksum": ck,
                               "headerless": hl, "orig_size": 0 if hl else len(d), "out": pack(o)})
    json.dump(out, open(os.path.join(HERE, "golden.json"), "w"), indent=0, separators=(",", ":"))
    print("stages", len(out["stages"]), "streams", len(out["streams"]))


if __name__ == "__main__":
    main()
