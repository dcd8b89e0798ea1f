"""oroch_amd -- MI355X (gfx950) implementation of Oroch's block integer-compression path.

The product is `liboroch_hip.so` (hand-written HIP kernels behind the C ABI of
`include/oroch_hip.h`) and the C++ host templates under `include/oroch/`.  This
package is the thin Python binding used by the tests and by `bench.py`; torch
is used only to own device memory and streams.
"""
from .capi import (  # noqa: F401
    BLOCK_VALUES,
    DTYPES,
    ENCODING_NAMES,
    GroupEncoder,
    IntegerArray,
    NOT_FOUND,
    OrochError,
    Pipe,
    bitfor_decode,
    bitfor_encode,
    bitpck_decode,
    bitpck_encode,
    bitpck_space,
    bitpfr_decode,
    bitpck_fetch,
    bitpfr_encode,
    dtype_code,
    group_bound,
    group_decode,
    group_decode_host,
    group_find,
    group_at,
    container_decode,
    container_decode_host,
    container_encode,
    container_encode_host,
    container_header,
    group_encode,
    varint_decode,
    varint_encode,
    varint_space,
    group_encode_host,
    group_select,
    group_stats,
    group_workspace,
    lib,
    option,
    set_option,
    library_path,
    profile_start,
    profile_stop,
    sequence_decode,
    sequence_encode,
    sequence_group_image,
    sequence_meta_decode,
    sequence_meta_encode,
    sequence_select,
)
from . import capi  # noqa: F401,E402

