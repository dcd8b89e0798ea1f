"""ctypes binding of liboroch_hip.so (the C ABI declared in include/oroch_hip.h).

Every function here goes to the HIP library; there is no CPU fallback.  If the
library is missing the import of this module fails.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
BLOCK_VALUES = 256
ENCODING_NAMES = ["naught", "normal", "varint", "varfor", "bitpck", "bitfor", "bitpfr"]
DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]

# numpy record matching struct oroch_block_desc
DESC_DTYPE = np.dtype([
    ("origin", "<u8"), ("dataspace", "<u4"), ("size", "<u4"), ("metaspace", "<u2"),
    ("noutliers", "<u2"), ("nvalues", "<u2"), ("encoding", "u1"), ("nbits", "u1"),
    ("index_encoding", "u1"), ("index_nbits", "u1"), ("value_encoding", "u1"),
    ("value_nbits", "u1"), ("reserved", "u1", (4,)),
])
assert DESC_DTYPE.itemsize == 32
# numpy record matching struct oroch_block_stats
STATS_DTYPE = np.dtype([("min", "<u8"), ("max", "<u8"), ("nvalues", "<u2"),
                        ("histogram", "<u2", (65,)), ("reserved", "<u2", (6,))])
assert STATS_DTYPE.itemsize == 160


class OrochError(RuntimeError):
    pass


def library_path():
    # OROCH_HIP_LIBRARY lets an experiment load another build of the same ABI
    return os.environ.get("OROCH_HIP_LIBRARY") or os.path.join(HERE, "liboroch_hip.so")


def _load():
    # torch bundles its own HIP runtime (same soname as the system one).  The
    # process must hold exactly one runtime, the one that owns torch's streams
    # and allocations, so torch is loaded first and this library binds to it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `make -C oroch_amd/csrc` (or "
            "__graft_entry__.build()); there is no CPU fallback" % path)
    L = C.CDLL(path)
    vp, sz, u32, u64, i = C.c_void_p, C.c_size_t, C.c_uint, C.c_uint64, C.c_int
    sig = {
        "oroch_hip_init": (i, [i]),
        "oroch_hip_strerror": (C.c_char_p, [i]),
        "oroch_hip_build_info": (C.c_char_p, []),
        "oroch_hip_release_host_buffers": (None, []),
        "oroch_bitpck_capacity": (sz, [u32]),
        "oroch_bitpck_space": (sz, [sz, u32]),
        "oroch_bitpck_encode": (i, [i, vp, sz, u32, vp, vp]),
        "oroch_bitpck_decode": (i, [i, vp, sz, u32, vp, vp]),
        "oroch_bitfor_encode": (i, [i, vp, sz, u64, u32, vp, vp]),
        "oroch_bitfor_decode": (i, [i, vp, sz, u64, u32, vp, vp]),
        "oroch_bitpfr_workspace": (sz, [sz]),
        "oroch_bitpfr_encode": (i, [i, vp, sz, u64, u32, vp, vp, vp, sz, vp, vp, sz, vp]),
        "oroch_bitpfr_decode": (i, [i, vp, sz, u64, u32, vp, vp, sz, vp, vp, sz, vp]),
        "oroch_bitpck_fetch": (i, [i, i, vp, sz, u64, u32, vp, sz, vp, vp]),
        "oroch_varint_workspace": (sz, [sz]),
        "oroch_varint_space": (i, [i, vp, sz, vp, sz, vp, vp]),
        "oroch_varint_encode": (i, [i, vp, sz, vp, sz, vp, sz, vp, vp]),
        "oroch_varint_decode": (i, [i, vp, sz, sz, vp, vp, sz, vp]),
        "oroch_varint_space_host": (i, [i, vp, sz, vp]),
        "oroch_varint_encode_host": (i, [i, vp, sz, vp, sz, vp]),
        "oroch_varint_decode_host": (i, [i, vp, sz, sz, vp, vp]),
        "oroch_bitpfr_patch": (i, [i, vp, sz, u64, u32, vp, vp, sz, vp, sz, vp]),
        "oroch_group_select": (i, [i, vp, sz, vp, vp]),
        "oroch_group_bound": (sz, [i, sz]),
        "oroch_group_workspace": (sz, [i, sz]),
        "oroch_group_encode": (i, [i, vp, sz, vp, sz, vp, vp, sz, vp, vp]),
        "oroch_group_decode": (i, [i, vp, vp, sz, vp, vp]),
        "oroch_group_encode_static": (i, [i, vp, sz, vp, sz, vp, vp]),
        "oroch_group_encode_static_max": (sz, [i]),
        "oroch_group_encode_host": (i, [i, vp, sz, vp, sz, vp, vp]),
        "oroch_group_decode_host": (i, [i, vp, vp, sz, vp]),
        "oroch_group_stats": (i, [i, vp, sz, vp, vp]),
        "oroch_group_select_host": (i, [i, vp, sz, vp]),
        "oroch_group_stats_host": (i, [i, vp, sz, vp]),
        "oroch_pack_host": (i, [i, i, vp, sz, u64, u32, vp]),
        "oroch_unpack_host": (i, [i, i, vp, sz, u64, u32, vp]),
        "oroch_bitpck_fetch_host": (i, [i, i, vp, u64, u32, vp, sz, vp]),
        "oroch_bitpfr_encode_host": (i, [i, vp, sz, u64, u32, vp, vp, vp, vp]),
        "oroch_bitpfr_decode_host": (i, [i, vp, sz, u64, u32, vp, vp, sz, vp]),
        "oroch_bitpfr_patch_host": (i, [i, vp, sz, u64, u32, vp, vp, sz]),
        "oroch_container_bound": (sz, [i, sz]),
        "oroch_container_encode": (i, [i, vp, sz, vp, sz, vp, sz, vp, vp]),
        "oroch_container_check": (i, [vp, sz]),
        "oroch_container_decode": (i, [vp, vp, sz, sz, vp, vp]),
        "oroch_container_encode_host": (i, [i, vp, sz, vp, sz, vp]),
        "oroch_container_decode_host": (i, [vp, sz, sz, sz, vp]),
        "oroch_group_find": (i, [i, vp, vp, sz, vp, sz, vp, vp]),
        "oroch_group_at": (i, [i, vp, vp, sz, vp, sz, vp, vp]),
        "oroch_array_create": (i, [i, vp]),
        "oroch_array_destroy": (None, [vp]),
        "oroch_array_size": (u64, [vp]),
        "oroch_array_clear": (i, [vp]),
        "oroch_array_append": (i, [vp, vp, sz]),
        "oroch_array_append_host": (i, [vp, vp, sz]),
        "oroch_array_insert": (i, [vp, u64, u64]),
        "oroch_array_find": (i, [vp, vp, sz, vp]),
        "oroch_array_at": (i, [vp, vp, sz, vp]),
        "oroch_array_read": (i, [vp, u64, u64, vp]),
        "oroch_array_image": (i, [vp, u64, vp, sz, vp]),
        "oroch_array_save_bound": (sz, [vp]),
        "oroch_array_save": (i, [vp, vp, sz, vp]),
        "oroch_array_load": (i, [vp, vp, sz]),
        "oroch_array_view": (i, [vp, vp, vp, vp, vp, vp, vp]),
        "oroch_pipe_create": (i, [sz, vp]),
        "oroch_pipe_destroy": (None, [vp]),
        "oroch_pipe_encode": (i, [vp, i, vp, sz, vp, sz, vp, vp]),
        "oroch_pipe_decode": (i, [vp, i, vp, vp, sz, vp]),
        "oroch_sequence_bound": (sz, [i, sz]),
        "oroch_sequence_select": (i, [i, vp, sz, vp, vp]),
        "oroch_sequence_encode": (i, [i, vp, sz, vp, vp, sz, vp, vp]),
        "oroch_sequence_decode": (i, [i, vp, sz, sz, vp, vp, vp, vp]),
        "oroch_sequence_select_host": (i, [i, vp, sz, vp]),
        "oroch_sequence_encode_host": (i, [i, vp, sz, vp, vp, sz, vp]),
        "oroch_sequence_decode_host": (i, [i, vp, sz, sz, vp, vp, vp]),
        "oroch_sequence_meta_encode": (i, [i, vp, vp, sz]),
        "oroch_sequence_meta_decode": (i, [i, vp, sz, vp]),
        "oroch_hip_profile_start": (i, [i]),
        "oroch_hip_profile_stop": (i, [vp, vp]),
        "oroch_hip_set_option": (i, [i, i]),
        "oroch_hip_get_option": (i, [i]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L, sorted(sig)


lib, EXPORTS = _load()


def _check(rc):
    if rc != 0:
        raise OrochError("oroch_hip error %d: %s" % (rc, lib.oroch_hip_strerror(rc).decode()))


def dtype_code(dt):
    """dtype code of a numpy dtype or a torch dtype."""
    if str(dt).startswith("torch."):
        dt = str(dt).replace("torch.", "")
    return [np.dtype(d).name for d in DTYPES].index(np.dtype(dt).name)


def _raw64(origin, code):
    bits = 8 << (code >> 1)
    return int(origin) & ((1 << bits) - 1)


def _stream():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _torch_dtype(code):
    import torch

    return [torch.int8, torch.uint8, torch.int16, torch.uint16, torch.int32, torch.uint32,
            torch.int64, torch.uint64][code]


def bitpck_space(n, nbits):
    return lib.oroch_bitpck_space(n, nbits)


def _empty_bytes(nbytes, device):
    import torch

    return torch.empty(max(16, (nbytes + 15) // 16 * 16), dtype=torch.uint8, device=device)


def bitpck_encode(x, nbits):
    """x: 1-D CUDA tensor; returns a uint8 CUDA tensor of bitpck_space bytes."""
    code = dtype_code(x.dtype)
    out = _empty_bytes(bitpck_space(x.numel(), nbits), x.device)
    _check(lib.oroch_bitpck_encode(code, x.data_ptr(), x.numel(), nbits, out.data_ptr(), _stream()))
    return out[:bitpck_space(x.numel(), nbits)]


def bitpck_decode(b, n, nbits, dtype):
    import torch

    code = dtype_code(dtype)
    out = torch.empty(n, dtype=_torch_dtype(code), device=b.device)
    _check(lib.oroch_bitpck_decode(code, b.data_ptr(), n, nbits, out.data_ptr(), _stream()))
    return out


def bitfor_encode(x, origin, nbits):
    code = dtype_code(x.dtype)
    out = _empty_bytes(bitpck_space(x.numel(), nbits), x.device)
    _check(lib.oroch_bitfor_encode(code, x.data_ptr(), x.numel(), _raw64(origin, code), nbits,
                                   out.data_ptr(), _stream()))
    return out[:bitpck_space(x.numel(), nbits)]


def bitfor_decode(b, n, origin, nbits, dtype):
    import torch

    code = dtype_code(dtype)
    out = torch.empty(n, dtype=_torch_dtype(code), device=b.device)
    _check(lib.oroch_bitfor_decode(code, b.data_ptr(), n, _raw64(origin, code), nbits,
                                   out.data_ptr(), _stream()))
    return out


def bitpfr_encode(x, origin, nbits, capacity=None):
    """Returns (packed bytes, gap-coded indices, high parts, nexc) as CUDA tensors / int."""
    import torch

    code = dtype_code(x.dtype)
    n = x.numel()
    cap = n if capacity is None else capacity
    out = _empty_bytes(bitpck_space(n, nbits), x.device)
    idx = torch.empty(max(cap, 1), dtype=torch.int64, device=x.device)
    val = torch.empty(max(cap, 1), dtype=torch.int64, device=x.device)
    nexc = torch.zeros(1, dtype=torch.int64, device=x.device)
    wsb = lib.oroch_bitpfr_workspace(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=x.device)
    _check(lib.oroch_bitpfr_encode(code, x.data_ptr(), n, _raw64(origin, code), nbits,
                                   out.data_ptr(), idx.data_ptr(), val.data_ptr(), cap,
                                   nexc.data_ptr(), ws.data_ptr(), wsb, _stream()))
    k = int(nexc.item())
    return out[:bitpck_space(n, nbits)], idx[:min(k, cap)], val[:min(k, cap)], k


def bitpfr_decode(b, n, origin, nbits, idx, val, dtype):
    import torch

    code = dtype_code(dtype)
    out = torch.empty(n, dtype=_torch_dtype(code), device=b.device)
    wsb = lib.oroch_bitpfr_workspace(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=b.device)
    _check(lib.oroch_bitpfr_decode(code, b.data_ptr(), n, _raw64(origin, code), nbits,
                                   idx.data_ptr(), val.data_ptr(), idx.numel(), out.data_ptr(),
                                   ws.data_ptr(), wsb, _stream()))
    return out


def bitpck_fetch(b, n, nbits, index, dtype, origin=None):
    """Values number index[i] of a packed sequence (bitfor when an origin is given)."""
    import torch

    code = dtype_code(dtype)
    out = torch.empty(index.numel(), dtype=_torch_dtype(code), device=b.device)
    _check(lib.oroch_bitpck_fetch(code, 0 if origin is None else 1, b.data_ptr(), n,
                                  0 if origin is None else _raw64(origin, code), nbits, index.data_ptr(),
                                  index.numel(), out.data_ptr(), _stream()))
    return out


def varint_space(x):
    """Bytes of the varint coding of a 1-D CUDA tensor (varint.h:124-131)."""
    import torch

    code = dtype_code(x.dtype)
    wsb = lib.oroch_varint_workspace(x.numel())
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
    out = C.c_uint64(0)
    _check(lib.oroch_varint_space(code, x.data_ptr(), x.numel(), ws.data_ptr(), wsb, C.byref(out), _stream()))
    return out.value


def varint_encode(x):
    import torch

    code = dtype_code(x.dtype)
    n = x.numel()
    worst = n * ((x.element_size() * 8 + 6) // 7)
    out = torch.empty(max(worst, 16), dtype=torch.uint8, device=x.device)
    wsb = lib.oroch_varint_workspace(n)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
    total = C.c_uint64(0)
    _check(lib.oroch_varint_encode(code, x.data_ptr(), n, out.data_ptr(), worst, ws.data_ptr(), wsb,
                                   C.byref(total), _stream()))
    return out[:total.value]


def varint_decode(b, n, dtype):
    import torch

    code = dtype_code(dtype)
    out = torch.zeros(n, dtype=_torch_dtype(code), device=b.device)
    wsb = lib.oroch_varint_workspace(b.numel())
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=b.device)
    _check(lib.oroch_varint_decode(code, b.data_ptr(), b.numel(), n, out.data_ptr(), ws.data_ptr(), wsb,
                                   _stream()))
    return out


# ---- host buffers through an overlapped pipeline ------------------------------------------


class Pipe:
    """oroch_pipe_*: encode / decode of host arrays in chunks, copies and kernels overlapped.
    Buffers may be numpy arrays (pageable) or pinned torch tensors viewed as numpy."""

    def __init__(self, chunk_bytes=0):
        self._h = C.c_void_p()
        _check(lib.oroch_pipe_create(chunk_bytes, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            lib.oroch_pipe_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

    def encode(self, a, out=None, offsets=None):
        """a: 1-D numpy array.  Returns (stream bytes, offsets) as numpy views of out / offsets."""
        code = dtype_code(a.dtype)
        n = a.size
        bound = lib.oroch_group_bound(code, n)
        if out is None:
            out = np.empty(bound, np.uint8)
        if offsets is None:
            offsets = np.empty((n + BLOCK_VALUES - 1) // BLOCK_VALUES + 1, np.uint64)
        total = C.c_uint64(0)
        _check(lib.oroch_pipe_encode(self._h, code, a.ctypes.data, n, out.ctypes.data, out.size,
                                     offsets.ctypes.data, C.byref(total)))
        return out[:total.value], offsets

    def decode(self, stream, offsets, n, dtype, out=None):
        code = dtype_code(dtype)
        if out is None:
            out = np.empty(n, DTYPES[code])
        _check(lib.oroch_pipe_decode(self._h, code, stream.ctypes.data, offsets.ctypes.data, n, out.ctypes.data))
        return out


# ---- one group of any length ------------------------------------------------------------

SEQUENCE_DESC = np.dtype([("origin", "<u8"), ("dataspace", "<u8"), ("noutliers", "<u8"), ("nvalues", "<u8"),
                          ("metaspace", "<u4"), ("encoding", "u1"), ("nbits", "u1"),
                          ("index_encoding", "u1"), ("index_nbits", "u1"), ("value_encoding", "u1"),
                          ("value_nbits", "u1"), ("reserved", "u1", (6,))])
assert SEQUENCE_DESC.itemsize == 48


def sequence_select(x):
    """integer_codec<T>::select on a 1-D CUDA tensor of any length -> descriptor (numpy record)."""
    d = np.zeros(1, SEQUENCE_DESC)
    _check(lib.oroch_sequence_select(dtype_code(x.dtype), x.data_ptr(), x.numel(), d.ctypes.data, _stream()))
    return d


def sequence_encode(x, desc, capacity=None):
    """The data part of x coded as desc says (uint8 CUDA tensor)."""
    code = dtype_code(x.dtype)
    cap = int(desc["dataspace"][0]) if capacity is None else capacity
    out = _empty_bytes(cap, x.device)
    total = C.c_uint64(0)
    _check(lib.oroch_sequence_encode(code, x.data_ptr(), x.numel(), desc.ctypes.data, out.data_ptr(), cap,
                                     C.byref(total), _stream()))
    return out[:total.value]


def sequence_decode(b, n, desc, dtype):
    """n values from a data part; returns (values, bytes consumed)."""
    import torch

    code = dtype_code(dtype)
    out = torch.empty(n, dtype=_torch_dtype(code), device=b.device)
    used = C.c_uint64(0)
    _check(lib.oroch_sequence_decode(code, b.data_ptr(), b.numel(), n, desc.ctypes.data, out.data_ptr(),
                                     C.byref(used), _stream()))
    return out, used.value


def sequence_meta_encode(desc, dtype):
    out = np.zeros(40, np.uint8)
    rc = lib.oroch_sequence_meta_encode(dtype_code(dtype), desc.ctypes.data, out.ctypes.data, out.size)
    if rc < 0:
        _check(rc)
    return out[:rc].copy()


def sequence_meta_decode(b, dtype):
    """Descriptor and number of metadata bytes at the start of a group image (numpy uint8)."""
    b = np.ascontiguousarray(b, np.uint8)
    d = np.zeros(1, SEQUENCE_DESC)
    rc = lib.oroch_sequence_meta_decode(dtype_code(dtype), b.ctypes.data, b.size, d.ctypes.data)
    if rc < 0:
        _check(rc)
    return d, rc


def sequence_group_image(x):
    """integer_group<T>::encode (integer_group.h:44-60, aligned) of a sequence of any
    length: [metadata][zero pad to 8][data], as numpy bytes."""
    d = sequence_select(x)
    meta = sequence_meta_encode(d, x.dtype)
    data = sequence_encode(x, d).cpu().numpy()
    at = (meta.size + 7) & ~7
    out = np.zeros(at + data.size, np.uint8)
    out[:meta.size] = meta
    out[at:] = data
    return out


KERNEL_NAMES = ["select", "scan", "encode", "decode", "pack", "unpack", "except"]
OPTIONS = {"encode_path": 0, "select_one_block": 1, "encode_one_block": 2, "decode_one_block": 3,
           "find_one_block": 4}


def set_option(name, value):
    """Sets a run-time switch (OROCH_OPT_*) and returns its previous value."""
    rc = lib.oroch_hip_set_option(OPTIONS[name], int(value))
    if rc < 0:
        _check(rc)
    return rc


class option:
    """`with option("decode_one_block", 1): ...` restores the switch afterwards."""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)

    def __exit__(self, *exc):
        set_option(self.name, self.old)


def profile_start(max_records=4096):
    _check(lib.oroch_hip_profile_start(max_records))


def profile_stop():
    """{kernel name: (total ms, launches)} for the launches since profile_start."""
    ms = (C.c_double * len(KERNEL_NAMES))()
    cnt = (C.c_uint * len(KERNEL_NAMES))()
    _check(lib.oroch_hip_profile_stop(ms, cnt))
    return {k: (ms[i], cnt[i]) for i, k in enumerate(KERNEL_NAMES) if cnt[i]}


def group_bound(dtype, n):
    return lib.oroch_group_bound(dtype_code(dtype), n)


def group_workspace(dtype, n):
    return lib.oroch_group_workspace(dtype_code(dtype), n)


def group_select(x):
    """Per-block selection records as a numpy structured array (DESC_DTYPE)."""
    import torch

    code = dtype_code(x.dtype)
    n = x.numel()
    nb = (n + BLOCK_VALUES - 1) // BLOCK_VALUES
    desc = torch.zeros(max(nb, 1) * 32, dtype=torch.uint8, device=x.device)
    _check(lib.oroch_group_select(code, x.data_ptr(), n, desc.data_ptr(), _stream()))
    torch.cuda.synchronize()
    return desc.cpu().numpy().view(DESC_DTYPE)[:nb]


def group_stats(x):
    """Per-block statistics records as a numpy structured array (STATS_DTYPE)."""
    import torch

    code = dtype_code(x.dtype)
    n = x.numel()
    nb = (n + BLOCK_VALUES - 1) // BLOCK_VALUES
    out = torch.zeros(max(nb, 1) * STATS_DTYPE.itemsize, dtype=torch.uint8, device=x.device)
    _check(lib.oroch_group_stats(code, x.data_ptr(), n, out.data_ptr(), _stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy().view(STATS_DTYPE)[:nb]


class GroupEncoder:
    """Pre-allocated buffers for repeated oroch_group_encode calls on one shape."""

    def __init__(self, dtype, n, device):
        import torch

        self.code = dtype_code(dtype)
        self.n = n
        self.nblocks = (n + BLOCK_VALUES - 1) // BLOCK_VALUES
        self.bound = lib.oroch_group_bound(self.code, n)
        self.wsb = lib.oroch_group_workspace(self.code, n)
        self.out = torch.empty(self.bound, dtype=torch.uint8, device=device)
        self.offsets = torch.empty(self.nblocks + 1, dtype=torch.int64, device=device)
        self.ws = torch.empty(max(self.wsb, 16), dtype=torch.uint8, device=device)

    def encode(self, x, want_total=False):
        total = C.c_uint64(0)
        _check(lib.oroch_group_encode(self.code, x.data_ptr(), self.n, self.out.data_ptr(),
                                      self.bound, self.offsets.data_ptr(), self.ws.data_ptr(),
                                      self.wsb, C.byref(total) if want_total else None,
                                      _stream()))
        return total.value if want_total else None


def group_encode(x):
    """Returns (stream bytes CUDA uint8 tensor, offsets CUDA int64 tensor)."""
    enc = GroupEncoder(x.dtype, x.numel(), x.device)
    total = enc.encode(x, want_total=True)
    return enc.out[:total], enc.offsets


def group_decode(stream, offsets, n, dtype):
    import torch

    code = dtype_code(dtype)
    out = torch.empty(n, dtype=_torch_dtype(code), device=stream.device)
    _check(lib.oroch_group_decode(code, stream.data_ptr(), offsets.data_ptr(), n, out.data_ptr(),
                                  _stream()))
    return out


def group_encode_host(a):
    """numpy in, numpy out, through the pinned-staging host entry point."""
    a = np.ascontiguousarray(a)
    code = dtype_code(a.dtype)
    n = a.size
    nb = (n + BLOCK_VALUES - 1) // BLOCK_VALUES
    bound = lib.oroch_group_bound(code, n)
    out = np.empty(bound, np.uint8)
    offs = np.zeros(nb + 1, np.uint64)
    total = C.c_uint64(0)
    _check(lib.oroch_group_encode_host(code, a.ctypes.data, n, out.ctypes.data, bound,
                                       offs.ctypes.data, C.byref(total)))
    return out[:total.value].copy(), offs


def group_decode_host(stream, offs, n, dtype):
    stream = np.ascontiguousarray(stream)
    offs = np.ascontiguousarray(offs, np.uint64)
    out = np.empty(n, dtype)
    _check(lib.oroch_group_decode_host(dtype_code(dtype), stream.ctypes.data, offs.ctypes.data, n,
                                       out.ctypes.data))
    return out


# ---- serialised container ------------------------------------------------------------

HEADER_DTYPE = np.dtype([("magic", "S8"), ("dtype", "<u4"), ("block_values", "<u4"), ("nvalues", "<u8"),
                         ("nblocks", "<u8"), ("table_offset", "<u8"), ("stream_offset", "<u8"),
                         ("stream_bytes", "<u8"), ("reserved", "<u8")])
assert HEADER_DTYPE.itemsize == 64


def container_encode(x):
    """x: 1-D CUDA tensor -> uint8 CUDA tensor holding the whole container."""
    import torch

    code = dtype_code(x.dtype)
    n = x.numel()
    bound = lib.oroch_container_bound(code, n)
    out = torch.empty(bound, dtype=torch.uint8, device=x.device)
    wsb = lib.oroch_group_workspace(code, n)
    ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=x.device)
    total = C.c_uint64(0)
    _check(lib.oroch_container_encode(code, x.data_ptr(), n, out.data_ptr(), bound, ws.data_ptr(), wsb,
                                      C.byref(total), _stream()))
    return out[:total.value]


def container_header(c):
    """Header record (numpy, HEADER_DTYPE) of a container held in a CUDA tensor or numpy array."""
    head = c[:64].cpu().numpy() if hasattr(c, "cpu") else np.ascontiguousarray(c[:64])
    h = head.view(HEADER_DTYPE)[0]
    nbytes = c.numel() if hasattr(c, "numel") else c.size
    _check(lib.oroch_container_check(head.ctypes.data, nbytes))
    return h


def container_decode(c, first_block=0, nblocks=None):
    """Decodes a block range of a container held in a CUDA uint8 tensor."""
    import torch

    head = c[:64].cpu().numpy()
    _check(lib.oroch_container_check(head.ctypes.data, c.numel()))
    h = head.view(HEADER_DTYPE)[0]
    if nblocks is None:
        nblocks = int(h["nblocks"]) - first_block
    lo = first_block * BLOCK_VALUES
    hi = min((first_block + nblocks) * BLOCK_VALUES, int(h["nvalues"]))
    out = torch.empty(max(hi - lo, 0), dtype=_torch_dtype(int(h["dtype"])), device=c.device)
    _check(lib.oroch_container_decode(c.data_ptr(), head.ctypes.data, first_block, nblocks, out.data_ptr(),
                                      _stream()))
    return out


def container_encode_host(a):
    a = np.ascontiguousarray(a)
    code = dtype_code(a.dtype)
    bound = lib.oroch_container_bound(code, a.size)
    out = np.empty(bound, np.uint8)
    total = C.c_uint64(0)
    _check(lib.oroch_container_encode_host(code, a.ctypes.data, a.size, out.ctypes.data, bound, C.byref(total)))
    return out[:total.value].copy()


def container_decode_host(c, first_block=0, nblocks=None):
    c = np.ascontiguousarray(c)
    _check(lib.oroch_container_check(c.ctypes.data, c.size))
    h = c[:64].view(HEADER_DTYPE)[0]
    if nblocks is None:
        nblocks = int(h["nblocks"]) - first_block
    lo = first_block * BLOCK_VALUES
    hi = min((first_block + nblocks) * BLOCK_VALUES, int(h["nvalues"]))
    out = np.empty(max(hi - lo, 0), DTYPES[int(h["dtype"])])
    _check(lib.oroch_container_decode_host(c.ctypes.data, c.size, first_block, nblocks, out.ctypes.data))
    return out


# ---- searching, random access, the growable array -----------------------------------------

NOT_FOUND = (1 << 64) - 1


def _raw_needles(values, code, device):
    """Values of the element type -> their raw bits zero-extended to 64 (CUDA int64 tensor)."""
    import torch

    a = np.atleast_1d(np.asarray(values)).astype(DTYPES[code])
    raw = a.view(np.dtype(DTYPES[code]).str.replace("i", "u")).astype(np.uint64)
    return torch.from_numpy(raw.view(np.int64)).to(device)


def group_find(stream, offsets, n, dtype, values):
    """First position of each value in the encoded array (numpy uint64, NOT_FOUND if absent)."""
    import torch

    code = dtype_code(dtype)
    needles = _raw_needles(values, code, stream.device)
    out = torch.empty(needles.numel(), dtype=torch.int64, device=stream.device)
    _check(lib.oroch_group_find(code, stream.data_ptr(), offsets.data_ptr(), n, needles.data_ptr(),
                                needles.numel(), out.data_ptr(), _stream()))
    return out.cpu().numpy().view(np.uint64)


def group_at(stream, offsets, n, dtype, index):
    """Values at the given positions (CUDA tensor of the element type)."""
    import torch

    code = dtype_code(dtype)
    idx = torch.as_tensor(np.asarray(index, np.uint64).view(np.int64)).to(stream.device)
    out = torch.empty(idx.numel(), dtype=_torch_dtype(code), device=stream.device)
    _check(lib.oroch_group_at(code, stream.data_ptr(), offsets.data_ptr(), n, idx.data_ptr(),
                              idx.numel(), out.data_ptr(), _stream()))
    return out


class IntegerArray:
    """oroch_array_* behind the names of the reference's integer_array<T>
    (integer_array.h:150-259), plus the bulk forms."""

    def __init__(self, dtype):
        self.dtype = np.dtype(dtype)
        self.code = dtype_code(self.dtype)
        self._h = C.c_void_p()
        _check(lib.oroch_array_create(self.code, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.oroch_array_destroy(self._h)
                self._h = None
        except Exception:  # interpreter shutdown: the library may be gone already
            pass

    def __len__(self):
        return lib.oroch_array_size(self._h)

    size = __len__

    def empty(self):
        return len(self) == 0

    def clear(self):
        _check(lib.oroch_array_clear(self._h))

    def append(self, values):
        """numpy array (host) or CUDA tensor (device)."""
        if hasattr(values, "data_ptr"):
            assert dtype_code(values.dtype) == self.code and values.is_contiguous()
            _check(lib.oroch_array_append(self._h, values.data_ptr(), values.numel()))
        else:
            a = np.ascontiguousarray(values, self.dtype)
            _check(lib.oroch_array_append_host(self._h, a.ctypes.data, a.size))

    def insert(self, index, value):
        _check(lib.oroch_array_insert(self._h, index, _raw64(value, self.code)))

    def find(self, values):
        """One value -> its first position or NOT_FOUND; a sequence -> numpy uint64 array."""
        a = np.atleast_1d(np.asarray(values)).astype(self.dtype)
        raw = np.ascontiguousarray(a.view(self.dtype.str.replace("i", "u")).astype(np.uint64))
        out = np.empty(raw.size, np.uint64)
        _check(lib.oroch_array_find(self._h, raw.ctypes.data, raw.size, out.ctypes.data))
        return int(out[0]) if np.ndim(values) == 0 else out

    def at(self, index):
        idx = np.ascontiguousarray(np.atleast_1d(index), np.uint64)
        out = np.empty(idx.size, self.dtype)
        _check(lib.oroch_array_at(self._h, idx.ctypes.data, idx.size, out.ctypes.data))
        return out[0] if np.ndim(index) == 0 else out

    __getitem__ = at

    def read(self, first=0, count=None):
        if count is None:
            count = len(self) - first
        out = np.empty(count, self.dtype)
        _check(lib.oroch_array_read(self._h, first, count, out.ctypes.data))
        return out

    def save(self):
        """The array as a serialised container (numpy uint8)."""
        out = np.empty(lib.oroch_array_save_bound(self._h), np.uint8)
        total = C.c_uint64(0)
        _check(lib.oroch_array_save(self._h, out.ctypes.data, out.size, C.byref(total)))
        return out[:total.value].copy()

    def load(self, container):
        c = np.ascontiguousarray(container, np.uint8)
        _check(lib.oroch_array_load(self._h, c.ctypes.data, c.size))

    def view(self):
        """(stream, offsets, tail): the encoded groups as CUDA tensors that alias the
        array's device memory (uint8 / int64; valid until the next modifying call;
        None when there is no group yet) and the raw tail as a numpy array."""
        import torch

        st, of, tl = C.c_void_p(), C.c_void_p(), C.c_void_p()
        nb, sb, nt = C.c_uint64(), C.c_uint64(), C.c_uint64()
        _check(lib.oroch_array_view(self._h, C.byref(st), C.byref(of), C.byref(nb), C.byref(sb),
                                    C.byref(tl), C.byref(nt)))
        tail = np.empty(nt.value, self.dtype)
        if nt.value:
            C.memmove(tail.ctypes.data, tl.value, tail.nbytes)
        if nb.value == 0:
            return None, None, tail

        class _Alias:
            def __init__(self, ptr, count, typestr):
                self.__cuda_array_interface__ = {"shape": (count,), "typestr": typestr,
                                                 "data": (ptr, False), "version": 2}

        stream = torch.as_tensor(_Alias(st.value, sb.value, "|u1"), device="cuda")
        offsets = torch.as_tensor(_Alias(of.value, nb.value + 1, "<i8"), device="cuda")
        return stream, offsets, tail
