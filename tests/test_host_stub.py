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
                           os.path.join(ROOT, "kanzi-cpp_amd", "host", "kanzi_amd.cpp"), obj, "-L" + ora, "-lknz_oracle",
                           "-Wl,-rpath," + ora, "-lpthread"])
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "host_mirror_test.cpp"), lib, "-Wl,-rpath," + str(tmp_path), "-L" + ora,
                           "-lknz_oracle", "-Wl,-rpath," + ora])
    return lib, exe


def test_host_layer_against_stub_device(tmp_path):
    lib, exe = build_stub(tmp_path)
    env = dict(os.environ, KNZ_TEST_KANZI_LIB=lib, KNZ_TEST_HOST_MIRROR_EXE=exe)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_host_api.py"), "-m", "gpu", "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "not threads_share_the_device"], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
