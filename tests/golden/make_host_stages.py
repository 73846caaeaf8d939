#!/usr/bin/env python3
"""Generates tests/golden/host_stages.json: what the UNMODIFIED reference's TEXT (both encodings) and UTF transforms make of a set of
inputs (oracle/_ref, TransformFactory with the entropy codec that selects the encoding): applied or refused, length and md5 of the output.

    make -C oracle ref && python tests/golden/make_host_stages.py

Data only (input specs + digests of the reference's output)."""
import hashlib, json, os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import knzlib, vectors  # noqa: E402

def main():
    R = knzlib.Ref()
    out = []
    for spec in vectors.HOST_STAGE_INPUTS:
        d = vectors.make(spec)
        rec = {"input": list(spec), "input_md5": hashlib.md5(d).hexdigest(), "stages": {}}
        for name, stage, ent, cap in (("text1", "TEXT", "FPAQ", len(d)), ("text2", "TEXT", "ANS0", len(d)), ("utf", "UTF", "", len(d) + 8192)):
            _, o, sk = R.forward(stage, d, dst_cap=cap, entropy=ent or None)
            applied = 0 if (sk & 0x80) else 1
            rec["stages"][name] = {"applied": applied, "len": len(o) if applied else 0, "md5": hashlib.md5(o).hexdigest() if applied else ""}
        out.append(rec)
        print(rec, flush=True)
    json.dump(out, open(os.path.join(HERE, "host_stages.json"), "w"), indent=1)

if __name__ == "__main__":
    main()
