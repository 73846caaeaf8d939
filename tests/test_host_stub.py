"""Host-layer logic on the CPU: kanzi-cpp_amd/host/kanzi_amd.cpp (stream classes, batching, page-locked staging slots and worker
thread, header parsing, seek/tell, the reference's C API) is linked against tests/stub/knz_hip_stub.c -- a stand-in for the
device library built on the oracle -- and driven by the same tests the GPU box runs (tests/cpp/host_mirror_test.cpp and
tests/test_gpu_host_api.py). What is under test is everything above the C ABI; the kernels below it are covered by the
`-m gpu` suite (real device) and tests/test_emu_kernels.py (emulated). Test infrastructure only: the product library is never
built this way."""
import os
import subprocess
import sys

import knzlib

ROOT = knzlib.ROOT


def build_stub(tmp_path):
    knzlib.ensure_oracle()
    obj = str(tmp_path / "stub.o")
    lib = str(tmp_path / "libkanzi_amd_stub.so")
    exe = str(tmp_path / "host_mirror_test_stub")
    ora = os.path.join(ROOT, "oracle")
    subprocess.check_call(["gcc", "-O1", "-std=c99", "-fPIC", "-Wall", "-c", os.path.join(ROOT, "tests", "stub", "knz_hip_stub.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", lib,
                           os.path.join(ROOT, "kanzi-cpp_amd", "host", "kanzi_amd.cpp"), os.path.join(ROOT, "kanzi-cpp_amd", "host", "text_codec.cpp"),
                           "-I" + os.path.join(ROOT, "kanzi-cpp_amd", "host"), obj, "-L" + ora, "-lknz_oracle",
                           "-Wl,-rpath," + ora, "-lpthread"])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), lib, "-Wl,-rpath," + str(tmp_path), "-L" + ora,
                           "-lknz_oracle", "-Wl,-rpath," + ora])
    cli = str(tmp_path / "kanzi_amd_cli_stub")
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", cli,
                           os.path.join(ROOT, "kanzi-cpp_amd", "host", "kanzi_cli.cpp"), lib, "-Wl,-rpath," + str(tmp_path), "-L" + ora,
                           "-lknz_oracle", "-Wl,-rpath," + ora])
    return lib, exe, cli


def test_host_layer_against_stub_device(tmp_path):
    lib, exe, cli = build_stub(tmp_path)
    env = dict(os.environ, KNZ_TEST_KANZI_LIB=lib, KNZ_TEST_HOST_MIRROR_EXE=exe, KNZ_TEST_DEVICES="0,1", KNZ_STUB_DEVICES="2", KNZ_TEST_CLI=cli)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_host_api.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "not threads_share_the_device and not bench_line_contract"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout


def test_cli_interoperates_with_the_reference_cli(tmp_path):
    """kanzi-cpp_amd/host/kanzi_cli.cpp (option names of src/app/Kanzi.cpp) against the reference's own `kanzi` binary
    (oracle/_ref/kanzi): same flags -> the same .knz bytes, each tool decodes the other's files, --from/--to select the
    same blocks. Runs on the CPU with the stand-in device library; the GPU suite repeats it with the real one."""
    import pytest
    ref = knzlib.REF_BIN
    if knzlib.ensure_ref() is None or not os.path.exists(ref):
        pytest.skip("reference build not available")
    _, _, cli = build_stub(tmp_path)
    run_cli_interop(cli, ref, tmp_path)


def run_cli_interop(cli, ref, tmp_path):
    import vectors
    data = vectors.make(("mixed", 5 * 65536 + 4321, 9))
    src = str(tmp_path / "in.bin")
    open(src, "wb").write(data)
    for args in (["-t", "BWT+MTFT+ZRLT", "-e", "ANS0", "-b", "64k"], ["-t", "LZX", "-e", "HUFFMAN", "-b", "65536", "-x"],
                 ["-l", "1", "-b", "1m"], ["--transform=RLT", "--entropy=FPAQ", "--block=32k", "-x64"]):
        a, b = str(tmp_path / "a.knz"), str(tmp_path / "b.knz")
        subprocess.check_call([ref, "-c", "-i", src, "-o", a, "-f", "-j", "1", "-v", "0"] + args)
        subprocess.check_call([cli, "-c", "-i", src, "-o", b, "-f", "-j", "1"] + args, stderr=subprocess.DEVNULL)
        assert open(a, "rb").read() == open(b, "rb").read(), args
        oa, ob = str(tmp_path / "a.out"), str(tmp_path / "b.out")
        subprocess.check_call([ref, "-d", "-i", b, "-o", oa, "-f", "-j", "1", "-v", "0"])
        subprocess.check_call([cli, "-d", "-i", a, "-o", ob, "-f"], stderr=subprocess.DEVNULL)
        assert open(oa, "rb").read() == data and open(ob, "rb").read() == data, args
    subprocess.check_call([ref, "-d", "-i", a, "-o", oa, "-f", "-j", "1", "-v", "0", "--from=2", "--to=4"])
    subprocess.check_call([cli, "-d", "-i", a, "-o", ob, "-f", "--from=2", "--to=4"], stderr=subprocess.DEVNULL)
    assert open(oa, "rb").read() == open(ob, "rb").read() == data[32768:3 * 32768]


_EIGHT_DEVICE_DRIVER = r'''
import hashlib, importlib, json, os, sys, time
sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
import knzlib, vectors
knzlib.load_pkg()
kz = importlib.import_module("kanzi_amd.kanzi")
kz.LIB_PATH = sys.argv[2]
bs, nblocks = 16384, 127
data = vectors.make(("mixed", bs * (nblocks - 1) + 5000, 77))
path = sys.argv[3]
t0 = time.perf_counter()
c = kz.Compressor(path, "NONE", "NONE", bs, 1)
for off in range(0, len(data), bs):
    c.compress(data[off:off + bs])
c.close()
t1 = time.perf_counter()
d = kz.Decompressor(path, buffer_size=bs, jobs=1)
out = bytearray()
while True:
    chunk = d.decompress(bs)
    out += chunk
    if len(chunk) < bs:
        break
d.close()
t2 = time.perf_counter()
assert bytes(out) == data
print(json.dumps({"enc_s": t1 - t0, "dec_s": t2 - t1, "md5": hashlib.md5(open(path, "rb").read()).hexdigest(), "blocks": nblocks}))
'''


def test_eight_device_lanes_scale_on_the_stub_device_model(tmp_path):
    """SURVEY.md 8(e) without the hardware: the stand-in device library models EIGHT devices, each of which runs one
    block call at a time and takes a fixed time per input byte (slept, not computed), and a 127-block job goes through the stream
    classes with one lane per device (KNZ_DEVICES=0,...,7, one block per batch). Asserted: (a) the 8-lane file is byte-identical with
    the one-device file and decodes to the input, (b) compress and decompress each take at most 1/6 of the one-device wall time
    (the block-count ceiling is 127 / 16 = 7.9). What this measures is the host layer's dispatch -- lanes filled in turn, runs
    appended in order, no lane waiting for another device -- which is the only scaling evidence obtainable on a box without GPUs;
    `test_two_or_more_physical_devices` (tests/test_gpu_host_api.py) is the same path on real devices."""
    import json
    lib, _, _ = build_stub(tmp_path)
    drv = str(tmp_path / "drv8.py")
    open(drv, "w").write(_EIGHT_DEVICE_DRIVER)
    def run(name, devs):
        env = dict(os.environ, KNZ_STUB_DEVICES="8", KNZ_STUB_NS_PER_BYTE="2442", KNZ_DEVICES=devs, KNZ_BATCH_BLOCKS="1")   # 40 ms per 16 KiB block
        r = subprocess.run([sys.executable, drv, ROOT, lib, str(tmp_path / (name + ".knz"))], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return json.loads(r.stdout.strip().splitlines()[-1])
    one = run("one", "0")
    assert one["enc_s"] > 127 * 0.040 * 0.95 and one["dec_s"] > 127 * 0.040 * 0.95, one          # (the model is what takes the time)
    # (wall times of sleeping threads: on a machine that is busy with something else a run may come out late -- the best of up to three)
    eight = None
    for attempt in range(3):
        e = run("eight", "0,1,2,3,4,5,6,7")
        assert one["md5"] == e["md5"]
        if eight is None:
            eight = e
        else:
            eight = dict(e, enc_s=min(e["enc_s"], eight["enc_s"]), dec_s=min(e["dec_s"], eight["dec_s"]))
        if eight["enc_s"] <= one["enc_s"] / 6.0 and eight["dec_s"] <= one["dec_s"] / 6.0:
            break
    assert eight["enc_s"] <= one["enc_s"] / 6.0, (one, eight)
    assert eight["dec_s"] <= one["dec_s"] / 6.0, (one, eight)
    print("stub device model, 127 blocks: one device %.3f / %.3f s, eight devices %.3f / %.3f s (x%.2f / x%.2f)" % (
        one["enc_s"], one["dec_s"], eight["enc_s"], eight["dec_s"], one["enc_s"] / eight["enc_s"], one["dec_s"] / eight["dec_s"]))
