#!/usr/bin/env python3
"""Generates tests/golden/golden_full.json: md5 + length of the UNMODIFIED reference's .knz (oracle/_ref) for the
BASELINE.json configurations at their own block sizes (vectors.FULL_CASES), 64 MiB inputs, -j 1, and for the long-common-prefix
inputs of vectors.HARD_CASES at 8 MiB / 32 MiB blocks (copies, periods, DNA; table- and image-shaped blocks), and for single blocks of
256 MiB and 1 GiB (vectors.BIG_CASES; the 1 GiB suffix sort takes the reference a few minutes and about 6 GB).

    make -C oracle ref && python tests/golden/make_golden_full.py [config ...]

With config names given only those records are (re)generated; the others are kept as they are.

Data only (input specs + digests of the reference's output); nothing of the reference's sources is stored.
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import knzlib  # noqa: E402
import vectors  # noqa: E402


def main():
    R = knzlib.Ref()
    out = []
    only = set(sys.argv[1:])
    cache = {}
    path = os.path.join(HERE, "golden_full.json")
    old = {str(r["config"]): r for r in json.load(open(path))} if only and os.path.exists(path) else {}
    for cfg, spec, t, e, bs in vectors.FULL_CASES + vectors.HARD_CASES + vectors.BIG_CASES:
        if only and str(cfg) not in only and str(cfg) in old:
            out.append(old[str(cfg)])
            continue
        d = cache[spec] if spec in cache else vectors.make(spec)
        cache.clear(); cache[spec] = d                     # (consecutive cases on the same input: one generation)
        rc, o = R.compress(d, t, e, bs, jobs=1, orig_size=len(d))
        assert rc == 0, (cfg, rc)
        rc, back = R.decompress(o, len(d))
        rec = {"config": cfg, "input": list(spec), "input_md5": hashlib.md5(d).hexdigest(), "transform": t, "entropy": e,
               "block": bs, "orig_size": len(d), "out": {"len": len(o), "md5": hashlib.md5(o).hexdigest()}}
        if rc != 0 and cfg == "big:bwt_1g":
            rec["ref_decode_error"] = rc               # the reference cannot read this stream of its own (see vectors.BIG_CASES)
        else:
            assert rc == 0 and back == d, cfg
        out.append(rec)
        print(out[-1], flush=True)
    json.dump(out, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
