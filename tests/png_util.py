"""Minimal PNG writer for the tests (stdlib zlib): every colour type, 8 / 16 bit, a chosen scan-line filter per row."""
import struct
import zlib

import numpy as np


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def _filter_row(ft, cur, up, bpp):
    cur, up = cur.astype(np.int32), up.astype(np.int32)
    a = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
    c = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]])
    if ft == 0:
        pred = 0
    elif ft == 1:
        pred = a
    elif ft == 2:
        pred = up
    elif ft == 3:
        pred = (a + up) >> 1
    else:
        p = a + up - c
        pa, pb, pc = np.abs(p - a), np.abs(p - up), np.abs(p - c)
        pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, up, c))
    return ((cur - pred) & 0xff).astype(np.uint8)


def write_png(path, samples, ctype, depth=8, filters=None, palette=None, trns=None, interlace=0):
    """samples: (h, w, channels) array of 8- or 16-bit samples in PNG channel order for `ctype`"""
    h, w, ch = samples.shape
    if depth == 16:
        raw = samples.astype(">u2").tobytes()
        rows = np.frombuffer(raw, np.uint8).reshape(h, w * ch * 2)
    else:
        rows = samples.astype(np.uint8).reshape(h, w * ch)
    bpp = ch * depth // 8
    out, prev = bytearray(), np.zeros(rows.shape[1], np.uint8)
    for y in range(h):
        ft = (filters[y % len(filters)] if filters else 0)
        out.append(ft)
        out += _filter_row(ft, rows[y], prev, bpp).tobytes()
        prev = rows[y]
    png = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, interlace))
    if palette is not None:
        png += _chunk(b"PLTE", np.asarray(palette, np.uint8).tobytes())
    if trns is not None:
        png += _chunk(b"tRNS", np.asarray(trns, np.uint8).tobytes())
    comp = zlib.compress(bytes(out), 6)
    half = len(comp) // 2
    png += _chunk(b"IDAT", comp[:half]) + _chunk(b"IDAT", comp[half:]) + _chunk(b"IEND", b"")   # two IDAT chunks on purpose
    with open(path, "wb") as f:
        f.write(png)
