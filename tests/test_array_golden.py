"""The compressed array against what the reference's own integer_array did with
the same script (tests/golden/array_v1.npz, written by make_array_golden.py from
the reference's headers): positions found by find() and the metadata lines of
group_info().

Without a GPU the fixture pins the checker itself: a plain numpy model of the
script reproduces the reference's answers, and the oracle's selection for every
full group reproduces the reference's metadata lines.  On the GPU the array does.
"""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
DTYPES = [np.int8, np.uint8, np.int16, np.uint16, np.int32, np.uint32, np.int64, np.uint64]
NOT_FOUND = (1 << 64) - 1


def load_cases():
    z = np.load(os.path.join(HERE, "golden", "array_v1.npz"), allow_pickle=False)
    doc = json.loads(bytes(z["cases"]).decode())
    out = []
    for i, c in enumerate(doc["cases"]):
        dt = np.dtype(DTYPES[c["dtype"]])
        out.append({
            "dt": dt,
            "values": z["a%d_values" % i].view(dt),
            "ins_pos": z["a%d_ins_pos" % i],
            "ins_val": z["a%d_ins_val" % i].view(dt),
            "taken": z["a%d_taken" % i].astype(bool),
            "needles": z["a%d_needles" % i].view(dt),
            "found": z["a%d_found" % i],
            "info": bytes(z["a%d_info" % i]).decode("latin-1"),
        })
    return out


def info_line(dt, encoding, origin_raw, nbits):
    """One line of group_info (integer_codec.h:219-224); an 8-bit origin goes
    through the stream as a character."""
    origin = np.array([origin_raw], np.uint64).astype("u%d" % dt.itemsize).view(dt)[0]
    shown = chr(int(origin) & 0xff) if dt.itemsize == 1 else str(int(origin))
    return "encoding: %d, origin: %s, nbits: %d\n" % (encoding, shown, nbits)


def replay(c):
    model = c["values"].copy()
    for pos, val, ok in zip(c["ins_pos"], c["ins_val"], c["taken"]):
        if ok:
            model = np.insert(model, int(pos), val)
    return model


def test_fixture_pins_the_model_and_the_oracle(oracle):
    cases = load_cases()
    assert len(cases) == 24
    assert sum(int(c["taken"].sum()) for c in cases) > 100
    for c in cases:
        model = replay(c)
        want = np.array([np.flatnonzero(model == v)[0] if (model == v).any() else NOT_FOUND
                         for v in c["needles"]], np.uint64)
        assert np.array_equal(want, c["found"]), c["dt"]
        text = ""
        for g in range(model.size // 256):
            m = oracle.select(model[g * 256:(g + 1) * 256])
            text += info_line(c["dt"], m.encoding, m.origin, m.nbits)
        assert text == c["info"], c["dt"]


@pytest.mark.gpu
def test_array_against_the_reference_scripts(oracle):
    import torch

    assert torch.cuda.is_available(), "these tests need the MI355X"
    import ctypes as C

    import oroch_amd as hip

    assert hip.lib.oroch_hip_init(0) == 0
    for c in load_cases():
        dt = c["dt"]
        arr = hip.IntegerArray(dt)
        # the reference inserted the values at the end one by one; the first 300 go
        # that way here too, the rest in bulk (the same groups either way)
        head = min(300, c["values"].size)
        for i in range(head):
            arr.insert(i, int(c["values"][i]))
        arr.append(c["values"][head:])
        for pos, val, ok in zip(c["ins_pos"], c["ins_val"], c["taken"]):
            if ok:
                arr.insert(int(pos), int(val))
        model = replay(c)
        assert len(arr) == model.size
        assert np.array_equal(arr.find(c["needles"]), c["found"]), dt
        for v, w in list(zip(c["needles"], c["found"]))[::5]:
            assert arr.find(v) == int(w), (dt, v)
        assert np.array_equal(arr.read(), model)
        # group_info: the metadata of every group, from its image
        text = ""
        image = np.zeros(dt.itemsize * 256 + 64, np.uint8)
        for g in range(model.size // 256):
            nbytes = C.c_uint64(0)
            rc = hip.lib.oroch_array_image(arr._h, g, image.ctypes.data, image.size, C.byref(nbytes))
            assert rc == 0
            m, _ = oracle.meta_decode(image[:nbytes.value], dt)
            text += info_line(dt, m.encoding, m.origin, m.nbits)
        assert text == c["info"], dt
        # an insertion the reference's range check turns away although it lies
        # inside the array goes through here
        refused = [(int(p), v) for p, v, ok in zip(c["ins_pos"], c["ins_val"], c["taken"])
                   if not ok and int(p) <= model.size]
        for pos, val in refused[:1]:
            arr.insert(pos, int(val))
            assert np.array_equal(arr.read(), np.insert(model, pos, val))
