"""FLAC reading through the library's host-side decoder (csrc/flac.hip): ``decode_flac(bytes) -> (float32 [T, ch], fs)`` with
the scaling of ``soundfile.read`` (integers / 2**(bits-1))."""
import ctypes

import numpy as np

from . import _lib


def flac_streaminfo(data, path="<bytes>"):
    lib = _lib.load()
    info = (ctypes.c_int64 * 6)()
    buf = ctypes.create_string_buffer(bytes(data[:1 << 16]), min(len(data), 1 << 16))
    if lib.urse_flac_info(ctypes.addressof(buf), len(buf), ctypes.addressof(info)) != 0:
        raise ValueError("%s: %s" % (path, lib.urse_last_error().decode()))
    return dict(fs=int(info[0]), channels=int(info[1]), bits=int(info[2]), total_samples=int(info[3]), min_block=int(info[4]),
                max_block=int(info[5]))


def decode_flac(data, path="<bytes>"):
    lib = _lib.load()
    si = flac_streaminfo(data, path)
    raw = np.frombuffer(data, dtype=np.uint8)
    cap = si["total_samples"] if si["total_samples"] > 0 else max(1, len(data) * 16 // max(1, si["channels"]))
    out = np.empty((cap, si["channels"]), dtype=np.int32)
    n = ctypes.c_int64()
    rc = lib.urse_flac_decode(raw.ctypes.data, len(raw), out.ctypes.data, cap, ctypes.addressof(n))
    if rc != 0:
        raise ValueError("%s: %s" % (path, lib.urse_last_error().decode()))
    x = out[:n.value].astype(np.float32) / np.float32(1 << (si["bits"] - 1))
    return x, si["fs"]
