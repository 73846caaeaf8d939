"""MI355X-native kanzi block pipeline (transform + entropy + block framing).

The package directory is named after the reference ("kanzi-cpp_amd"); because of the hyphen it
is imported through importlib (see __graft_entry__.load_package / tests/knzlib.load_pkg) under the
module name ``kanzi_amd``. Compute lives in csrc/ (HIP, gfx950) behind the C ABI declared in
include/knz_hip.h; the Python here only mirrors the reference's ctypes wrapper.
"""
__version__ = "0.1.0"
