"""ctypes binding of the device C-ABI (include/knz_hip.h, built into kanzi-cpp_amd/libknz_hip.so).

This module is the Python-side counterpart of the reference's src/api/kanzi_c_api.py for the
device layer: it only marshals pointers and sizes. There is no CPU fallback: if the library is
missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libknz_hip.so")

E_NONE, E_HUFFMAN, E_FPAQ, E_ANS0, E_ANS1 = 0, 1, 2, 5, 8
ENTROPY_IDS = {"NONE": 0, "HUFFMAN": 1, "FPAQ": 2, "ANS0": 5, "ANS1": 8}
TRANSFORM_IDS = {"NONE": 0, "BWT": 1, "LZ": 3, "RLT": 5, "ZRLT": 6, "MTFT": 7, "RANK": 8, "SRT": 13, "LZX": 16, "TIMESTAMP": 64}

SYMBOLS = [
    "knz_hip_device_count", "knz_hip_create", "knz_hip_destroy", "knz_hip_last_error", "knz_hip_encode_bound",
    "knz_hip_encode_blocks", "knz_hip_decode_blocks", "knz_hip_entropy_encode", "knz_hip_entropy_decode",
    "knz_hip_transform_forward", "knz_hip_transform_inverse", "knz_hip_malloc", "knz_hip_free",
    "knz_hip_memcpy_h2d", "knz_hip_memcpy_d2h", "knz_hip_sync", "knz_hip_memcpy_h2d_async", "knz_hip_memcpy_d2h_async", "knz_hip_copy_wait", "knz_hip_host_alloc", "knz_hip_host_free", "knz_hip_set_profiling", "knz_hip_get_kernel_times",
    "knz_hip_tune", "knz_hip_shift_bits", "knz_hip_encode_block_hosted", "knz_hip_decode_block_hosted",
    "knz_hip_entropy_decode_v", "knz_hip_transform_inverse_v",
]


class Params(C.Structure):
    _fields_ = [("transform_type", C.c_uint64), ("entropy_type", C.c_int32), ("block_size", C.c_int32),
                ("checksum_bits", C.c_int32), ("jobs", C.c_int32), ("bs_version", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("ms", C.c_float), ("launches", C.c_uint64)]


class KnzError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("knz_hip error %d: %s" % (code, msg))
        self.code = code


def transform_type(names):
    """TransformFactory::getType (transform/TransformFactory.hpp:100-137): 6 bits per stage, NONE dropped."""
    res, shift = 0, 42
    toks = names.upper().split("+")
    if len(toks) > 8:
        raise ValueError("Only 8 transforms allowed")
    if len(toks) == 1:
        return TRANSFORM_IDS[toks[0]] << 42
    for t in toks:
        v = TRANSFORM_IDS[t]
        if v:
            res |= v << shift
            shift -= 6
    return res


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libknz_hip.so not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
        L = C.CDLL(LIB_PATH)
        vp, u8p, sz = C.c_void_p, C.c_void_p, C.c_size_t
        L.knz_hip_device_count.argtypes = [C.POINTER(C.c_int)]
        L.knz_hip_create.argtypes = [C.c_int, vp, C.POINTER(vp)]
        L.knz_hip_destroy.argtypes = [vp]; L.knz_hip_destroy.restype = None
        L.knz_hip_last_error.argtypes = [vp]; L.knz_hip_last_error.restype = C.c_char_p
        L.knz_hip_encode_bound.argtypes = [C.POINTER(Params), sz]; L.knz_hip_encode_bound.restype = sz
        L.knz_hip_encode_blocks.argtypes = [vp, C.POINTER(Params), u8p, sz, C.c_char_p, C.c_uint32, C.c_int64, C.c_int,
                                            u8p, sz, C.POINTER(C.c_uint64)]
        L.knz_hip_decode_blocks.argtypes = [vp, C.POINTER(Params), u8p, C.c_uint64, C.c_uint64, C.c_int64, u8p, sz,
                                            C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)]
        L.knz_hip_entropy_encode.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint32, u8p, sz, C.POINTER(C.c_uint64)]
        L.knz_hip_entropy_decode.argtypes = [vp, C.c_int, C.c_char_p, C.c_uint64, C.c_uint64, u8p, C.c_uint32,
                                             C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.knz_hip_entropy_decode_v.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_uint64, C.c_uint64, u8p, C.c_uint32,
                                               C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
        L.knz_hip_transform_inverse_v.argtypes = [vp, C.c_int, C.c_int, C.c_char_p, C.c_int32, u8p, C.c_int32,
                                                  C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.knz_hip_transform_forward.argtypes = [vp, C.c_int, C.c_char_p, C.c_int32, u8p, C.c_int32, C.c_int,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.knz_hip_transform_inverse.argtypes = [vp, C.c_int, C.c_char_p, C.c_int32, u8p, C.c_int32,
                                                C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        L.knz_hip_malloc.argtypes = [vp, sz, C.POINTER(vp)]
        L.knz_hip_free.argtypes = [vp, vp]
        L.knz_hip_memcpy_h2d.argtypes = [vp, vp, C.c_char_p, sz]
        L.knz_hip_memcpy_d2h.argtypes = [vp, vp, vp, sz]
        L.knz_hip_sync.argtypes = [vp]
        L.knz_hip_set_profiling.argtypes = [vp, C.c_int]
        L.knz_hip_get_kernel_times.argtypes = [vp, C.POINTER(KernelTime), C.c_int]
        L.knz_hip_tune.argtypes = [C.c_char_p, C.c_int]
        L.knz_hip_shift_bits.argtypes = [vp, u8p, C.c_uint64, C.c_uint32, u8p]
        _lib = L
    return _lib


class Context:
    """One device context (one per GPU / per process rank)."""

    def __init__(self, device=0, stream=None):
        L = lib()
        h = C.c_void_p()
        rc = L.knz_hip_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != 0:
            raise KnzError(rc, "cannot create device context (no GPU visible?)")
        self.h = h
        self.L = L

    def close(self):
        if self.h:
            self.L.knz_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise KnzError(rc, self.L.knz_hip_last_error(self.h).decode())

    # ---- device memory
    def malloc(self, n):
        p = C.c_void_p()
        self._chk(self.L.knz_hip_malloc(self.h, n, C.byref(p)))
        return p.value

    def free(self, p):
        self._chk(self.L.knz_hip_free(self.h, C.c_void_p(p)))

    def h2d(self, dptr, data):
        self._chk(self.L.knz_hip_memcpy_h2d(self.h, C.c_void_p(dptr), data, len(data)))

    def d2h(self, dptr, n):
        buf = (C.c_uint8 * max(1, n))()
        self._chk(self.L.knz_hip_memcpy_d2h(self.h, buf, C.c_void_p(dptr), n))
        return C.string_at(buf, n)

    def sync(self):
        self._chk(self.L.knz_hip_sync(self.h))

    # ---- batch API on device pointers
    def params(self, transform, entropy, block_size, checksum=0, jobs=1, bs_version=0):
        p = Params()
        p.transform_type = transform_type(transform) if isinstance(transform, str) else transform
        p.entropy_type = ENTROPY_IDS[entropy.upper()] if isinstance(entropy, str) else entropy
        p.block_size = block_size
        p.checksum_bits = checksum
        p.jobs = jobs
        p.bs_version = bs_version
        return p

    def encode_bound(self, p, n):
        return self.L.knz_hip_encode_bound(C.byref(p), n)

    def encode_blocks(self, p, d_in, n, d_out, out_cap, prologue=b"", prologue_bits=0, first_block=0, finish=1):
        bits = C.c_uint64(0)
        self._chk(self.L.knz_hip_encode_blocks(self.h, C.byref(p), C.c_void_p(d_in), n, prologue, prologue_bits,
                                               first_block, finish, C.c_void_p(d_out), out_cap, C.byref(bits)))
        return bits.value

    def decode_blocks(self, p, d_in, in_bits, start_bit, d_out, out_cap, max_blocks=0):
        ob, eb, nb = C.c_uint64(0), C.c_uint64(0), C.c_int64(0)
        self._chk(self.L.knz_hip_decode_blocks(self.h, C.byref(p), C.c_void_p(d_in), in_bits, start_bit, max_blocks,
                                               C.c_void_p(d_out), out_cap, C.byref(ob), C.byref(eb), C.byref(nb)))
        return ob.value, eb.value, nb.value

    # ---- per-stage API on host buffers
    def entropy_encode(self, entropy, data):
        e = ENTROPY_IDS[entropy.upper()]
        cap = 2 * len(data) + 65536
        out = (C.c_uint8 * cap)()
        bits = C.c_uint64(0)
        self._chk(self.L.knz_hip_entropy_encode(self.h, e, data, len(data), out, cap, C.byref(bits)))
        return C.string_at(out, (bits.value + 7) // 8), bits.value

    def entropy_decode(self, entropy, enc, n, start_bit=0, in_bits=None, bs_version=0):
        e = ENTROPY_IDS[entropy.upper()]
        out = (C.c_uint8 * max(1, n))()
        dec, used = C.c_int32(0), C.c_uint64(0)
        if in_bits is None:
            in_bits = 8 * len(enc)
        self._chk(self.L.knz_hip_entropy_decode_v(self.h, e, bs_version, enc, in_bits, start_bit, out, n, C.byref(dec), C.byref(used)))
        return dec.value, C.string_at(out, n), used.value

    def transform_forward(self, transform, data, dst_cap, entropy=None):
        t = TRANSFORM_IDS[transform.upper()]
        out = (C.c_uint8 * (max(dst_cap, len(data)) + 2048))()
        ol, ok = C.c_int32(0), C.c_int32(0)
        e = ENTROPY_IDS[entropy.upper()] if entropy else -1
        self._chk(self.L.knz_hip_transform_forward(self.h, t, data, len(data), out, dst_cap, e, C.byref(ol), C.byref(ok)))
        return ok.value, C.string_at(out, ol.value)

    def transform_inverse(self, transform, data, dst_cap, bs_version=0):
        t = TRANSFORM_IDS[transform.upper()]
        out = (C.c_uint8 * (dst_cap + 64))()
        ol, ok = C.c_int32(0), C.c_int32(0)
        self._chk(self.L.knz_hip_transform_inverse_v(self.h, t, bs_version, data, len(data), out, dst_cap, C.byref(ol), C.byref(ok)))
        return ok.value, C.string_at(out, ol.value)

    # ---- profiling
    def set_profiling(self, on):
        self.L.knz_hip_set_profiling(self.h, 1 if on else 0)

    def kernel_times(self):
        arr = (KernelTime * 256)()
        n = self.L.knz_hip_get_kernel_times(self.h, arr, 256)
        return [(arr[i].name.decode(), arr[i].ms, arr[i].launches) for i in range(n)]
