"""A small PNG reader for the tests: 8-bit RGB, any of the five row filters, chunk CRCs checked.
Returns the pixels as 0xFFRRGGBB plus the filter type of every row."""
import zlib

import numpy as np


def decode_png(path):
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, W, H = 8, b"", 0, 0
    while pos < len(raw):
        n = int.from_bytes(raw[pos:pos + 4], "big")
        typ = raw[pos + 4:pos + 8]
        body = raw[pos + 8:pos + 8 + n]
        assert zlib.crc32(raw[pos + 4:pos + 8 + n]) == int.from_bytes(raw[pos + 8 + n:pos + 12 + n], "big")
        if typ == b"IHDR":
            W, H = int.from_bytes(body[0:4], "big"), int.from_bytes(body[4:8], "big")
            assert tuple(body[8:13]) == (8, 2, 0, 0, 0)  # 8 bits, RGB, deflate, adaptive filtering, no interlace
        elif typ == b"IDAT":
            idat += body
        pos += 12 + n
    data = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + 3 * W)
    filters = data[:, 0].copy()
    out = np.zeros((H, 3 * W), np.int64)
    prev = np.zeros(3 * W, np.int64)
    for y in range(H):
        f, row = int(filters[y]), data[y, 1:].astype(np.int64)
        cur = np.zeros(3 * W, np.int64)
        if f == 0:
            cur = row
        elif f == 2:
            cur = (row + prev) & 255
        else:
            for i in range(3 * W):
                a = cur[i - 3] if i >= 3 else 0
                b = prev[i]
                c = prev[i - 3] if i >= 3 else 0
                if f == 1:
                    pred = a
                elif f == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (row[i] + pred) & 255
        out[y] = cur
        prev = cur
    px = out.reshape(H * W, 3).astype(np.uint32)
    return (255 << 24) | (px[:, 0] << 16) | (px[:, 1] << 8) | px[:, 2], filters, (W, H), len(idat)
