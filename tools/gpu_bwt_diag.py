"""BWT forward on small lone blocks against the oracle for every round-0 key length the knob allows (developer diagnostic)."""
import importlib, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import knzlib, vectors
knzlib.load_pkg()
hipapi = importlib.import_module("kanzi_amd.hipapi")
hip = hipapi.Context(0)
ora = knzlib.Oracle()
rng = np.random.default_rng(1)
cases = {"zeros5000": bytes(5000), "zeros300": bytes(300), "zeros9000": bytes(9000), "runs": vectors.make(("runs", 300, 40)),
         "text8000": vectors.make(("text", 8000, 3)), "rand4000": rng.integers(0, 256, 4000, dtype=np.uint8).tobytes(),
         "two": bytes(2500) + bytes([1]) + bytes(2499), "zeros70000": bytes(70000)}
for name, d in cases.items():
    ok1, o1 = ora.forward("BWT", d, len(d) + 64, "ANS0")
    for packed in ("0", "1", "2", "3", "4", "5"):
        hipapi.lib().knz_hip_tune(b"bwt_nsym", int(packed))
        for rep in range(2):
            ok2, o2 = hip.transform_forward("BWT", d, len(d) + 64, "ANS0")
            same = bool(ok1) == bool(ok2) and o1 == o2
            first = next((i for i in range(min(len(o1), len(o2))) if o1[i] != o2[i]), -1)
            print(name, "nsym=" + packed, "rep", rep, "OK" if same else "MISMATCH at %d (%d vs %d bytes) hdr %s vs %s" % (first, len(o1), len(o2), o1[:6].hex(), o2[:6].hex()), flush=True)
