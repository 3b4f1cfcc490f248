"""Minimal PNG writer for the tests (stdlib zlib): every colour type and bit depth, Adam7 interlacing, a chosen scan-line filter per row."""
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


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]


def _pack_rows(samples, depth):
    """(h, w, ch) samples -> (h, row_bytes) uint8 scan lines (16-bit big-endian; 1 / 2 / 4 bit packed leftmost-high)"""
    h, w, ch = samples.shape
    if depth == 16:
        return np.frombuffer(samples.astype(">u2").tobytes(), np.uint8).reshape(h, w * ch * 2)
    if depth == 8:
        return samples.astype(np.uint8).reshape(h, w * ch)
    per = 8 // depth
    v = samples.astype(np.uint8).reshape(h, w * ch)
    pad = (-v.shape[1]) % per
    v = np.concatenate([v, np.zeros((h, pad), np.uint8)], axis=1).reshape(h, -1, per)
    out = np.zeros(v.shape[:2], np.uint16)
    for k in range(per):
        out |= v[:, :, k].astype(np.uint16) << (8 - depth * (k + 1))
    return out.astype(np.uint8)


def write_png(path, samples, ctype, depth=8, filters=None, palette=None, trns=None, interlace=0):
    """samples: (h, w, channels) array of samples (values below 2**depth) in PNG channel order for `ctype`; interlace=1: Adam7"""
    h, w, ch = samples.shape
    bpp = max(1, ch * depth // 8)
    out = bytearray()
    passes = [(0, 0, 1, 1)] if not interlace else ADAM7
    row_no = 0
    for (x0, y0, dx, dy) in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        rows = _pack_rows(sub, depth)
        prev = np.zeros(rows.shape[1], np.uint8)
        for y in range(rows.shape[0]):
            ft = (filters[row_no % len(filters)] if filters else 0)
            row_no += 1
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
