"""Compressor / Decompressor classes over the C API of libkanzi_amd.so.

Mirrors the reference's ctypes shim (src/api/kanzi_c_api.py:88-137 for the bindings,
src/api/kanzi.py for the two classes): same struct layouts, same call sequence
(init -> compress()/decompress() per block -> dispose). Only marshals data.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libkanzi_amd.so")

C_API_SYMBOLS = ["getCompressorVersion", "initCompressor", "compress", "disposeCompressor",
                 "getDecompressorVersion", "initDecompressor", "decompress", "disposeDecompressor"]


class cData(C.Structure):
    _fields_ = [("transform", C.c_char * 64), ("entropy", C.c_char * 16), ("blockSize", C.c_size_t),
                ("jobs", C.c_uint), ("checksum", C.c_int), ("headerless", C.c_int)]


class dData(C.Structure):
    _fields_ = [("bufferSize", C.c_size_t), ("jobs", C.c_uint), ("headerless", C.c_int), ("transform", C.c_char * 64),
                ("entropy", C.c_char * 16), ("blockSize", C.c_uint), ("originalSize", C.c_size_t), ("checksum", C.c_int),
                ("bsVersion", C.c_int)]


_lib = None
_libc = None


def lib():
    global _lib, _libc
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libkanzi_amd.so not built: run __graft_entry__.build()")
        L = C.CDLL(LIB_PATH)
        L.getCompressorVersion.restype = C.c_uint
        L.getDecompressorVersion.restype = C.c_uint
        L.initCompressor.argtypes = [C.POINTER(cData), C.c_void_p, C.POINTER(C.c_void_p)]
        L.compress.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.disposeCompressor.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.initDecompressor.argtypes = [C.POINTER(dData), C.c_void_p, C.POINTER(C.c_void_p)]
        L.decompress.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.disposeDecompressor.argtypes = [C.POINTER(C.c_void_p)]
        _lib = L
        _libc = C.CDLL(None)
        _libc.fopen.restype = C.c_void_p
        _libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
        _libc.fclose.argtypes = [C.c_void_p]
    return _lib


class _PyBuffer(C.Structure):
    """Py_buffer (CPython's object.h): lets compress() hand the caller's own memory to the library, whatever
    bytes-like object it lives in (bytes, bytearray, memoryview, numpy array; read-only included), without a copy."""
    _fields_ = [("buf", C.c_void_p), ("obj", C.py_object), ("len", C.c_ssize_t), ("itemsize", C.c_ssize_t), ("readonly", C.c_int),
                ("ndim", C.c_int), ("format", C.c_char_p), ("shape", C.POINTER(C.c_ssize_t)), ("strides", C.POINTER(C.c_ssize_t)),
                ("suboffsets", C.POINTER(C.c_ssize_t)), ("internal", C.c_void_p)]


C.pythonapi.PyObject_GetBuffer.argtypes = [C.py_object, C.POINTER(_PyBuffer), C.c_int]
C.pythonapi.PyObject_GetBuffer.restype = C.c_int
C.pythonapi.PyBuffer_Release.argtypes = [C.POINTER(_PyBuffer)]
C.pythonapi.PyBuffer_Release.restype = None
C.pythonapi.PyBytes_FromStringAndSize.argtypes = [C.c_char_p, C.c_ssize_t]
C.pythonapi.PyBytes_FromStringAndSize.restype = C.py_object
C.pythonapi.PyBytes_AsString.argtypes = [C.py_object]
C.pythonapi.PyBytes_AsString.restype = C.c_void_p


class KanziError(RuntimeError):
    def __init__(self, code, what):
        super().__init__("%s failed with kanzi error %d" % (what, code))
        self.code = code


def _name(v):
    """Codec names arrive as str or, as in the reference's shim (src/api/kanzi.py), as bytes."""
    if isinstance(v, (bytes, bytearray)):
        return bytes(v)
    if isinstance(v, str):
        return v.encode()
    raise TypeError("codec name must be str or bytes, not %s" % type(v).__name__)


class Compressor:
    """src/api/kanzi.py Compressor: same positional/keyword arguments, `compress(data)`, `close()`, context manager."""

    def __init__(self, path, transform="NONE", entropy="NONE", block_size=4 << 20, jobs=1, checksum=0, headerless=False):
        L = lib()
        tname, ename = _name(transform), _name(entropy)
        self._f = _libc.fopen(os.fsencode(path), b"wb")
        if not self._f:
            raise OSError("cannot open %s" % path)
        self.params = cData(tname, ename, block_size, jobs, checksum, 1 if headerless else 0)
        self._ctx = C.c_void_p()
        rc = L.initCompressor(C.byref(self.params), self._f, C.byref(self._ctx))
        if rc != 0:
            _libc.fclose(self._f)
            self._f = None
            raise KanziError(rc, "initCompressor")
        self.written = 0

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def compress(self, data):
        out = C.c_size_t(0)
        view = _PyBuffer()
        if C.pythonapi.PyObject_GetBuffer(data, C.byref(view), 0) != 0:     # PyBUF_SIMPLE: contiguous bytes
            raise TypeError("compress() needs a contiguous bytes-like object")
        try:
            rc = lib().compress(self._ctx, view.buf, view.len, C.byref(out))
        finally:
            C.pythonapi.PyBuffer_Release(C.byref(view))
        if rc != 0:
            raise KanziError(rc, "compress")
        self.written += out.value
        return out.value

    def close(self):
        if self._ctx:
            out = C.c_size_t(0)
            rc = lib().disposeCompressor(C.byref(self._ctx), C.byref(out))
            self.written += out.value
            self._ctx = C.c_void_p()
            _libc.fclose(self._f)
            self._f = None
            if rc != 0:
                raise KanziError(rc, "disposeCompressor")
        return self.written


class Decompressor:
    """src/api/kanzi.py Decompressor: `Decompressor(path, buffer_size, jobs, headerless, **headerless_params)` with the
    reference's parameter names (transform, entropy, blockSize, originalSize, checksum, bsVersion; the snake_case
    spellings are accepted too), `decompress_block(max_output)`, `close()`, context manager."""

    def __init__(self, path, buffer_size=4 << 20, jobs=1, headerless=False, transform="NONE", entropy="NONE", block_size=0,
                 original_size=0, checksum=0, bs_version=6, **ref_names):
        L = lib()
        block_size = ref_names.pop("blockSize", block_size)
        original_size = ref_names.pop("originalSize", original_size)
        bs_version = ref_names.pop("bsVersion", bs_version)
        if ref_names:
            raise TypeError("unexpected arguments: %s" % ", ".join(sorted(ref_names)))
        self._f = _libc.fopen(os.fsencode(path), b"rb")
        if not self._f:
            raise OSError("cannot open %s" % path)
        self.params = dData(buffer_size, jobs, 1 if headerless else 0, _name(transform), _name(entropy), block_size,
                            original_size, checksum, bs_version)
        self._ctx = C.c_void_p()
        rc = L.initDecompressor(C.byref(self.params), self._f, C.byref(self._ctx))
        if rc != 0:
            _libc.fclose(self._f)
            self._f = None
            raise KanziError(rc, "initDecompressor")
        self.buffer_size = buffer_size

    def __enter__(self):
        return self

    def __exit__(self, exc_type, exc, tb):
        self.close()

    def decompress_block(self, max_output):
        return self.decompress(max_output)

    def decompress(self, n):
        # the library writes straight into the bytes object handed back (no zero fill, no second copy); only a
        # short last block is cut to size
        res = C.pythonapi.PyBytes_FromStringAndSize(None, max(1, n))
        ins, outs = C.c_size_t(0), C.c_size_t(n)
        rc = lib().decompress(self._ctx, C.pythonapi.PyBytes_AsString(res), C.byref(ins), C.byref(outs))
        if rc != 0:
            raise KanziError(rc, "decompress")
        return res if outs.value == len(res) else res[:outs.value]

    def close(self):
        if self._ctx:
            rc = lib().disposeDecompressor(C.byref(self._ctx))
            self._ctx = C.c_void_p()
            _libc.fclose(self._f)
            self._f = None
            if rc != 0:
                raise KanziError(rc, "disposeDecompressor")
