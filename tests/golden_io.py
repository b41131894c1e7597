"""Reader for the GOLD1 fixture files written by oracle/gen_golden.cpp (data only)."""
import os
import struct

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_DT = {0: np.float32, 1: np.uint8, 2: np.int32, 3: np.uint16}


def load_gold(name):
    path = os.path.join(GOLDEN_DIR, name)
    with open(path, "rb") as f:
        buf = f.read()
    assert buf[:5] == b"GOLD1", path
    (n,) = struct.unpack_from("<I", buf, 5)
    off = 9
    out = {}
    for _ in range(n):
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4
        key = buf[off:off + ln].decode()
        off += ln
        dt, nb = struct.unpack_from("<IQ", buf, off)
        off += 12
        out[key] = np.frombuffer(buf, dtype=_DT[dt], count=nb // np.dtype(_DT[dt]).itemsize, offset=off).copy()
        off += nb
    return out
