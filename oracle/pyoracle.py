"""ctypes bindings for the test-side checkers.

TEST INFRASTRUCTURE ONLY: importable from ``tests/``, ``__graft_entry__.smoke``
and the ``cpu_baseline`` leg of ``bench.py``; the product package must never
import this module.

Two libraries with identical flat signatures are bound:

* ``oracle``  -- ``liboroch_oracle.so``, the C restatement (``oroch_oracle.c``)
* ``ref``     -- ``_ref/liboroch_ref.so``, the real reference headers behind
  ``ref_shim.cc`` (present only where it has been built from the reference
  tree; ``ref`` is ``None`` otherwise)
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

NAUGHT, NORMAL, VARINT, VARFOR, BITPCK, BITFOR, BITPFR = range(7)
ENCODING_NAMES = ["naught", "normal", "varint", "varfor", "bitpck", "bitfor", "bitpfr"]

DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]


def dtype_code(dt):
    dt = np.dtype(dt)
    return DTYPES.index(dt.type)


class Meta(C.Structure):
    """Flat image of encoding_metadata<T> (oroch_oracle.h: orc_meta)."""

    _fields_ = [
        ("encoding", C.c_uint32),
        ("nbits", C.c_uint32),
        ("origin", C.c_uint64),
        ("dataspace", C.c_uint64),
        ("metaspace", C.c_uint64),
        ("noutliers", C.c_uint64),
        ("index_encoding", C.c_uint32),
        ("index_nbits", C.c_uint32),
        ("value_encoding", C.c_uint32),
        ("value_nbits", C.c_uint32),
    ]

    def astuple(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)

    def __repr__(self):
        return "Meta(%s)" % ", ".join("%s=%d" % (f, getattr(self, f)) for f, _ in self._fields_)


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Lib:
    """One of the two checker libraries, with numpy-friendly wrappers."""

    def __init__(self, path, prefix):
        self.lib = C.CDLL(path)
        self.prefix = prefix
        f = self._f
        vp, sz, u32, u64 = C.c_void_p, C.c_size_t, C.c_uint, C.c_uint64
        f("bitpck_space", sz, [sz, sz])
        f("bitpck_encode", sz, [C.c_int, vp, sz, u32, vp])
        f("bitpck_decode", sz, [C.c_int, vp, sz, u32, vp])
        f("bitfor_encode", sz, [C.c_int, vp, sz, u64, u32, vp])
        f("bitfor_decode", sz, [C.c_int, vp, sz, u64, u32, vp])
        f("bitpfr_encode", sz, [C.c_int, vp, sz, u64, u32, vp, vp, vp, vp])
        f("bitpfr_decode", sz, [C.c_int, vp, sz, u64, u32, vp, vp, sz, vp])
        f("varint_encode", sz, [C.c_int, vp, sz, vp])
        f("varint_decode", sz, [C.c_int, vp, sz, vp])
        f("stats", None, [C.c_int, vp, sz, vp, vp, vp])
        f("select", None, [C.c_int, vp, sz, C.POINTER(Meta)])
        f("meta_encode", sz, [C.c_int, C.POINTER(Meta), vp])
        f("meta_decode", sz, [C.c_int, vp, C.POINTER(Meta)])
        f("codec_encode", sz, [C.c_int, vp, sz, C.POINTER(Meta), vp])
        f("codec_decode", sz, [C.c_int, vp, sz, C.POINTER(Meta), vp])
        f("group_encode", sz, [C.c_int, vp, sz, vp])
        f("blocks_encode", sz, [C.c_int, vp, sz, sz, sz, vp, vp])
        if prefix == "orc_":
            f("group_decode", sz, [C.c_int, vp, sz, vp])
            f("blocks_decode", None, [C.c_int, vp, vp, sz, sz, vp])
        else:
            f("group_decode", sz, [C.c_int, vp, sz, sz, vp])
            f("bench_blocks", None, [C.c_int, vp, sz, sz, C.c_int, vp])
            f("bench_whole", None, [C.c_int, vp, sz, C.c_int, u64, u32, vp])
            f("array_script", sz, [C.c_int, vp, sz, vp, vp, sz, vp, vp, sz, vp, vp, sz])

    def _f(self, name, restype, argtypes):
        fn = getattr(self.lib, self.prefix + name)
        fn.restype = restype
        fn.argtypes = argtypes
        setattr(self, "_" + name, fn)

    # -- helpers -----------------------------------------------------------
    @staticmethod
    def pck_space(n, nbits):
        if n == 0:
            return 0
        c = 128 // nbits
        return 16 * ((n + c - 1) // c)

    def bitpck_encode(self, a, nbits):
        a = np.ascontiguousarray(a)
        out = np.zeros(self.pck_space(a.size, nbits) + 16, np.uint8)
        k = self._bitpck_encode(dtype_code(a.dtype), _ptr(a), a.size, nbits, _ptr(out))
        return out[:k].copy()

    def bitpck_decode(self, b, n, nbits, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(16, np.uint8)]))
        out = np.zeros(n, dt)
        self._bitpck_decode(dtype_code(dt), _ptr(b), n, nbits, _ptr(out))
        return out

    def bitfor_encode(self, a, origin, nbits):
        a = np.ascontiguousarray(a)
        out = np.zeros(self.pck_space(a.size, nbits) + 16, np.uint8)
        k = self._bitfor_encode(dtype_code(a.dtype), _ptr(a), a.size, raw64(origin, a.dtype),
                                nbits, _ptr(out))
        return out[:k].copy()

    def bitfor_decode(self, b, n, origin, nbits, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(16, np.uint8)]))
        out = np.zeros(n, dt)
        self._bitfor_decode(dtype_code(dt), _ptr(b), n, raw64(origin, dt), nbits, _ptr(out))
        return out

    def bitpfr_encode(self, a, origin, nbits):
        a = np.ascontiguousarray(a)
        out = np.zeros(self.pck_space(a.size, nbits) + 16, np.uint8)
        idx = np.zeros(a.size + 1, np.uint64)
        val = np.zeros(a.size + 1, np.uint64)
        k = C.c_size_t(0)
        nb = self._bitpfr_encode(dtype_code(a.dtype), _ptr(a), a.size, raw64(origin, a.dtype),
                                 nbits, _ptr(out), _ptr(idx), _ptr(val), C.byref(k))
        return out[:nb].copy(), idx[:k.value].copy(), val[:k.value].copy()

    def bitpfr_decode(self, b, n, origin, nbits, idx, val, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(16, np.uint8)]))
        idx = np.ascontiguousarray(idx, np.uint64)
        val = np.ascontiguousarray(val, np.uint64)
        out = np.zeros(n, dt)
        self._bitpfr_decode(dtype_code(dt), _ptr(b), n, raw64(origin, dt), nbits, _ptr(idx),
                            _ptr(val), idx.size, _ptr(out))
        return out

    def varint_encode(self, a):
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size * 10 + 16, np.uint8)
        k = self._varint_encode(dtype_code(a.dtype), _ptr(a), a.size, _ptr(out))
        return out[:k].copy()

    def varint_decode(self, b, n, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(16, np.uint8)]))
        out = np.zeros(n, dt)
        self._varint_decode(dtype_code(dt), _ptr(b), n, _ptr(out))
        return out

    def stats(self, a):
        a = np.ascontiguousarray(a)
        mn, mx = C.c_uint64(0), C.c_uint64(0)
        hist = np.zeros(65, np.uint64)
        self._stats(dtype_code(a.dtype), _ptr(a), a.size, C.byref(mn), C.byref(mx), _ptr(hist))
        return mn.value, mx.value, hist

    def select(self, a):
        a = np.ascontiguousarray(a)
        m = Meta()
        self._select(dtype_code(a.dtype), _ptr(a), a.size, C.byref(m))
        return m

    def meta_encode(self, m, dt):
        out = np.zeros(64, np.uint8)
        k = self._meta_encode(dtype_code(dt), C.byref(m), _ptr(out))
        return out[:k].copy()

    def meta_decode(self, b, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(32, np.uint8)]))
        m = Meta()
        k = self._meta_decode(dtype_code(dt), _ptr(b), C.byref(m))
        return m, k

    def codec_encode(self, a, m):
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size * a.itemsize * 2 + 64, np.uint8)
        k = self._codec_encode(dtype_code(a.dtype), _ptr(a), a.size, C.byref(m), _ptr(out))
        return out[:k].copy()

    def codec_decode(self, b, n, m, dt):
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(32, np.uint8)]))
        out = np.zeros(n, dt)
        k = self._codec_decode(dtype_code(dt), _ptr(b), n, C.byref(m), _ptr(out))
        return out, k

    def group_encode(self, a):
        a = np.ascontiguousarray(a)
        out = np.zeros(a.size * a.itemsize + 64, np.uint8)
        k = self._group_encode(dtype_code(a.dtype), _ptr(a), a.size, _ptr(out))
        return out[:k].copy()

    def group_decode(self, b, n, dt):
        nb = b.size
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(32, np.uint8)]))
        out = np.zeros(n, dt)
        if self.prefix == "orc_":
            self._group_decode(dtype_code(dt), _ptr(b), n, _ptr(out))
        else:
            self._group_decode(dtype_code(dt), _ptr(b), nb + 32, n, _ptr(out))
        return out

    def blocks_encode(self, a, block=256, align=16):
        a = np.ascontiguousarray(a)
        nblocks = (a.size + block - 1) // block
        out = np.zeros(a.size * a.itemsize + 32 * nblocks + 64, np.uint8)
        offs = np.zeros(nblocks + 1, np.uint64)
        k = self._blocks_encode(dtype_code(a.dtype), _ptr(a), a.size, block, align, _ptr(out),
                                _ptr(offs))
        return out[:k].copy(), offs

    def blocks_decode(self, b, offs, n, dt, block=256):
        assert self.prefix == "orc_"
        b = np.ascontiguousarray(np.concatenate([b, np.zeros(32, np.uint8)]))
        offs = np.ascontiguousarray(offs, np.uint64)
        out = np.zeros(n, dt)
        self._blocks_decode(dtype_code(dt), _ptr(b), _ptr(offs), n, block, _ptr(out))
        return out

    def bench_blocks(self, a, block=256, threads=1):
        """Reference only: seconds for (select, encode, decode), bytes, ok."""
        a = np.ascontiguousarray(a)
        out = np.zeros(5, np.float64)
        self._bench_blocks(dtype_code(a.dtype), _ptr(a), a.size, block, threads, _ptr(out))
        return out


def _bench_whole(self, a, kind, origin, nbits):
    """Reference only: seconds to (encode, decode) the whole sequence with the
    caller's parameters (kind 0 bitpck, 1 bitfor, 2 bitpfr), and whether the
    values came back."""
    a = np.ascontiguousarray(a)
    out = np.zeros(3, np.float64)
    self._bench_whole(dtype_code(a.dtype), _ptr(a), a.size, kind, raw64(origin, a.dtype), nbits, _ptr(out))
    return out


Lib.bench_whole = _bench_whole


def ref_array_script(values, ins_pos, ins_val, needles):
    """The reference's integer_array: insert values one by one at the end, then
    ins_val[k] at ins_pos[k], then find() every needle.  Returns (positions as
    uint64, text of group_info(), which insertions the reference accepted)."""
    values = np.ascontiguousarray(values)
    dt = values.dtype
    ins_pos = np.ascontiguousarray(ins_pos, np.uint64)
    ins_val = np.ascontiguousarray(ins_val, dt)
    needles = np.ascontiguousarray(needles, dt)
    found = np.zeros(needles.size, np.uint64)
    taken = np.zeros(ins_pos.size, np.uint8)
    info = np.zeros(64 * (values.size // 256 + ins_pos.size + 2) + 64, np.uint8)
    k = ref._array_script(dtype_code(dt), _ptr(values), values.size, _ptr(ins_pos), _ptr(ins_val),
                          ins_pos.size, _ptr(taken), _ptr(needles), needles.size, _ptr(found), _ptr(info),
                          info.size)
    return found, bytes(info[:k]).decode("latin-1"), taken.astype(bool)


def raw64(v, dt):
    """Raw bit pattern of v in dtype dt, zero-extended to 64 bits."""
    dt = np.dtype(dt)
    u = np.dtype("u%d" % dt.itemsize)
    return int(np.array([v]).astype(dt).view(u)[0])


def _load(path, prefix):
    return Lib(path, prefix) if os.path.exists(path) else None


oracle = _load(os.path.join(HERE, "liboroch_oracle.so"), "orc_")
ref = _load(os.path.join(HERE, os.environ.get("OROCH_REF_LIBRARY", os.path.join("_ref", "liboroch_ref.so"))), "ref_")
