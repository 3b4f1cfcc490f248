"""Sanitizer runs of the host-side code, on the CPU (no GPU sanitizer exists on this pool):
  * libvfhip's file parsers (csrc/host_parsers.hip: PNG decoder, .cube parser, PNG-LUT slicer; csrc/host_jpeg.hip: JPEG decoder) built as plain C++ with
    -fsanitize=address,undefined and fed valid, truncated, bit-flipped and adversarial files — the answer may be an error
    code, never a sanitizer report, a crash or an exception escaping the C ABI;
  * oracle/metalref.c on awkward frame sizes in exactly sized heap buffers (tests/asan/oracle_harness.c);
  * oracle/gst114.c + metalref.c through their whole CPU suites (all golden vectors) with the sanitizer build of the oracle
    library loaded into python (LD_PRELOAD libasan).
Test infrastructure only; nothing here is on the product path."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

import png_util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN = ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:halt_on_error=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")


def have(tool):
    return subprocess.run(["which", tool], capture_output=True).returncode == 0


pytestmark = pytest.mark.skipif(not (have("gcc") and have("g++")), reason="needs gcc / g++ with libasan")


@pytest.fixture(scope="module")
def parsers(tmp_path_factory):
    exe = tmp_path_factory.mktemp("asan") / "parsers_harness"
    subprocess.check_call(["g++"] + SAN + ["-x", "c++", "-o", str(exe), os.path.join(ROOT, "tests", "asan", "parsers_harness.cpp"),
                                           os.path.join(ROOT, "gstreamer-metal_amd", "csrc", "host_parsers.hip"),
                                           os.path.join(ROOT, "gstreamer-metal_amd", "csrc", "host_jpeg.hip"), "-lz"])
    return str(exe)


def run(exe, files):
    r = subprocess.run([exe] + [str(f) for f in files], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, (r.stdout[-1500:], r.stderr[-3000:])
    return {ln.split(": rc ")[0]: int(ln.split(": rc ")[1].split()[0]) for ln in r.stdout.splitlines() if ": rc " in ln}


def chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)


def test_png_decoder_under_asan(parsers, tmp_path):
    rng = np.random.default_rng(7)
    files, good = [], []
    for k, (ctype, ch, depth) in enumerate([(0, 1, 8), (2, 3, 8), (4, 2, 8), (6, 4, 8), (6, 4, 16), (2, 3, 16)]):
        p = tmp_path / f"ok{k}.png"
        png_util.write_png(p, rng.integers(0, 256 if depth == 8 else 65536, (13, 17, ch)), ctype, depth, filters=[0, 1, 2, 3, 4])
        files.append(p); good.append(str(p))
    p = tmp_path / "pal.png"
    png_util.write_png(p, rng.integers(0, 5, (9, 11, 1)), 3, 8, palette=rng.integers(0, 256, (5, 3), dtype=np.uint8), trns=[0, 128, 255])
    files.append(p); good.append(str(p))
    for k, (ctype, ch, depth, il) in enumerate([(0, 1, 1, 0), (0, 1, 2, 1), (0, 1, 4, 1), (3, 1, 1, 1), (3, 1, 4, 0), (6, 4, 8, 1), (2, 3, 16, 1), (4, 2, 8, 1)]):
        p = tmp_path / f"ok_low{k}.png"                   # low bit depths and Adam7 (sizes that leave passes empty or one pixel wide)
        w, h = [(13, 17), (1, 1), (3, 2), (5, 9)][k % 4]
        png_util.write_png(p, rng.integers(0, min(1 << depth, 256), (h, w, ch)), ctype, depth, filters=[4, 3, 2, 1, 0],
                           palette=rng.integers(0, 256, (1 << depth, 3), dtype=np.uint8) if ctype == 3 else None, interlace=il)
        files.append(p); good.append(str(p))
    base = open(good[3], "rb").read()
    for n in list(range(0, 60)) + list(range(60, len(base), 7)):                    # every truncation point of the header, then strided
        q = tmp_path / f"trunc{n}.png"; q.write_bytes(base[:n]); files.append(q)
    for k in range(200):                                                             # random byte flips (CRCs are not checked: data reaches the decoder)
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 4))):
            b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
        q = tmp_path / f"flip{k}.png"; q.write_bytes(bytes(b)); files.append(q)
    for j, g in enumerate(good[7:]):                                                 # the same for the packed-sample / Adam7 files
        gb = open(g, "rb").read()
        for k in range(40):
            b = bytearray(gb)
            for _ in range(int(rng.integers(1, 4))):
                b[int(rng.integers(8, len(b)))] = int(rng.integers(0, 256))
            q = tmp_path / f"flip_low{j}_{k}.png"; q.write_bytes(bytes(b)); files.append(q)
        for n in range(8, len(gb), 5):
            q = tmp_path / f"trunc_low{j}_{n}.png"; q.write_bytes(gb[:n]); files.append(q)
    sig = b"\x89PNG\r\n\x1a\n"
    adversarial = {
        "huge.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 16384, 16384, 16, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""),
        "zero.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0, 5, 8, 6, 0, 0, 0)) + chunk(b"IEND", b""),
        "neg.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 0xFFFFFFFF, 0x80000000, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 64)) + chunk(b"IEND", b""),
        "lenlie.png": sig + struct.pack(">I", 0xFFFFFFF0) + b"IHDR" + b"\0" * 40,
        "shortidat.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 64, 64, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 100)) + chunk(b"IEND", b""),
        "longidat.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\1" * 5000)) + chunk(b"IEND", b""),
        "badfilter.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 2, 8, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\7abcd\5efgh")) + chunk(b"IEND", b""),
        "palidx.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 1, 8, 3, 0, 0, 0)) + chunk(b"PLTE", b"\1\2\3") + chunk(b"IDAT", zlib.compress(b"\0\0\1\2\xff")) + chunk(b"IEND", b""),
        "nopal.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 1, 8, 3, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0\0\0\0\0")) + chunk(b"IEND", b""),
        "interlaced_len.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\0" * 40)) + chunk(b"IEND", b""),   # 7 passes of 4x4 hold 23 bytes
        "interlace2.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 4, 4, 8, 0, 0, 0, 2)) + chunk(b"IDAT", zlib.compress(b"\0" * 20)) + chunk(b"IEND", b""),
        "depth1_len.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 9, 8, 1, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 16)) + chunk(b"IEND", b""),       # 9 pixels need 2 bytes per row
        "depth3.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 3, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 32)) + chunk(b"IEND", b""),
        "rgb_depth4.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 8, 4, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(b"\0" * 104)) + chunk(b"IEND", b""),
        "pal1_idx.png": sig + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 1, 1, 3, 0, 0, 0)) + chunk(b"PLTE", b"\1\2\3") + chunk(b"IDAT", zlib.compress(b"\0\x10")) + chunk(b"IEND", b""),   # index 1, one-entry palette
        "empty.png": b"", "sigonly.png": sig,
    }
    for name, data in adversarial.items():
        q = tmp_path / name; q.write_bytes(data); files.append(q)
    files.append(tmp_path / "missing.png")
    rc = run(parsers, files)
    assert all(rc[g] == 0 for g in good), {g: rc[g] for g in good}
    for name in adversarial:
        if name != "prog_refine_first.jpg":               # (DC refinement bits without a first scan decode to something; it must only be memory-safe)
            assert rc[str(tmp_path / name)] < 0, name
    assert rc[str(tmp_path / "missing.png")] < 0 and rc[str(tmp_path / "trunc0.png")] < 0 and rc[str(tmp_path / "trunc33.png")] < 0


def test_cube_and_png_lut_parsers_under_asan(parsers, tmp_path):
    rng = np.random.default_rng(8)
    n = 4
    body = "\n".join("%.4f %.4f %.4f" % tuple(rng.random(3)) for _ in range(n ** 3))
    cases = {
        "ok.cube": f"# c\nTITLE \"x\"\nLUT_3D_SIZE {n}\nDOMAIN_MIN 0 0 0\nDOMAIN_MAX 1 1 1\n{body}\n",
        "crlf.cube": f"LUT_3D_SIZE {n}\r\n" + body.replace("\n", "\r\n") + "\r\n",
        "short.cube": f"LUT_3D_SIZE {n}\n" + "\n".join(body.split("\n")[:-3]) + "\n",
        "nosize.cube": body + "\n",
        "size0.cube": "LUT_3D_SIZE 0\n0 0 0\n", "size1.cube": "LUT_3D_SIZE 1\n0 0 0\n", "size65.cube": "LUT_3D_SIZE 65\n",
        "sizeneg.cube": "LUT_3D_SIZE -7\n0 0 0\n", "sizehuge.cube": "LUT_3D_SIZE 99999999999999999999\n0 0 0\n", "sizetext.cube": "LUT_3D_SIZE abc\n",
        "extra.cube": f"LUT_3D_SIZE 2\n" + "0 0 0\n" * 40,
        "resize.cube": f"LUT_3D_SIZE 3\n" + "0 0 0\n" * 5 + "LUT_3D_SIZE 2\n" + "1 1 1\n" * 8,
        "longline.cube": "LUT_3D_SIZE 2\n" + "0.5 " * 4000 + "\n" + "0 0 0\n" * 8,
        "nan.cube": "LUT_3D_SIZE 2\n" + "nan inf -inf\n" * 8, "garbage.cube": "".join(chr(int(c)) for c in rng.integers(1, 127, 3000)),
        "binary.cube": None, "empty.cube": "",
    }
    files, want_ok = [], {"ok.cube", "crlf.cube", "extra.cube", "resize.cube", "nan.cube"}
    for name, text in cases.items():
        q = tmp_path / name
        q.write_bytes(bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) if text is None else text.encode())
        files.append(q)
    # PNG LUTs: a valid 4^3 LUT (2 slices per row), and pictures whose pixel count is no cube / whose layout cannot hold the slices
    png_util.write_png(tmp_path / "ok.lut.png", rng.integers(0, 256, (8, 8, 3)), 2)
    png_util.write_png(tmp_path / "odd.lut.png", rng.integers(0, 256, (5, 7, 3)), 2)
    png_util.write_png(tmp_path / "thin.lut.png", rng.integers(0, 256, (64, 1, 3)), 2)        # 64 = 4^3 pixels, but 1 column < one 4-pixel slice
    png_util.write_png(tmp_path / "wide.lut.png", rng.integers(0, 256, (1, 27, 3)), 2)        # 27 = 3^3 pixels in one row: 9 slices of 3 rows cannot fit
    files += [tmp_path / "ok.lut.png", tmp_path / "odd.lut.png", tmp_path / "thin.lut.png", tmp_path / "wide.lut.png", tmp_path / "missing.cube"]
    rc = run(parsers, files)
    for name in cases:
        if name == "longline.cube":
            continue                                       # a 16 kB line is read in 511-byte pieces: accepted or refused, never overrun
        assert (rc[str(tmp_path / name)] == 0) == (name in want_ok), (name, rc[str(tmp_path / name)])
    assert rc[str(tmp_path / "ok.lut.png")] == 0
    assert rc[str(tmp_path / "odd.lut.png")] < 0 and rc[str(tmp_path / "thin.lut.png")] < 0 and rc[str(tmp_path / "wide.lut.png")] < 0 and rc[str(tmp_path / "missing.cube")] < 0


def test_metalref_oracle_under_asan_on_awkward_sizes(tmp_path):
    exe = tmp_path / "oracle_harness"
    subprocess.check_call(["gcc"] + SAN + ["-ffp-contract=off", "-o", str(exe), os.path.join(ROOT, "tests", "asan", "oracle_harness.c"), os.path.join(ROOT, "oracle", "metalref.c"), "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=ENV, timeout=300)
    assert r.returncode == 0 and "no sanitizer report" in r.stdout and "runtime error" not in r.stderr, (r.stdout[-500:], r.stderr[-3000:])


def test_oracle_suites_under_asan(tmp_path):
    """every golden vector of oracle/gst114.c and the metalref self-checks, with the sanitizer build of the oracle library"""
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("libasan.so not found")
    so = tmp_path / "libvfhip_oracle_asan.so"
    subprocess.check_call(["gcc"] + SAN + ["-ffp-contract=off", "-fPIC", "-fopenmp", "-shared", "-o", str(so), os.path.join(ROOT, "oracle", "gst114.c"),
                                           os.path.join(ROOT, "oracle", "metalref.c"), "-lm"])
    env = dict(ENV, LD_PRELOAD=libasan, VFHIP_ORACLE_LIB=str(so), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_golden.py"),
                        os.path.join(ROOT, "tests", "test_metalref_cpu.py")], capture_output=True, text=True, env=env, timeout=1500, cwd=ROOT)
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stdout + r.stderr and "runtime error:" not in r.stdout + r.stderr, (r.stdout[-3000:], r.stderr[-3000:])
    assert " passed" in r.stdout


def test_jpeg_decoder_under_asan(parsers, tmp_path):
    """valid JPEGs of every supported kind (sequential and progressive), every truncation point of two small ones, 900 random byte flips (the entropy-coded data and the tables
    reach the decoder unchecked), hand-made adversarial headers: error codes are fine, sanitizer reports are not"""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(17)
    files, good = [], []
    pic = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    pic[10:20, 10:30] = (250, 20, 40)
    for k, kw in enumerate([dict(quality=75, subsampling=0), dict(quality=40, subsampling=1), dict(quality=90, subsampling=2), dict(quality=60, subsampling=2, optimize=True),
                            dict(quality=100, subsampling=0)]):
        p = tmp_path / f"ok{k}.jpg"
        Image.fromarray(pic).save(p, **kw)
        files.append(p); good.append(str(p))
    g = tmp_path / "grey.jpg"
    Image.fromarray(pic[..., 0], mode="L").save(g, quality=70)
    files.append(g); good.append(str(g))
    for k, kw in enumerate([dict(quality=75, subsampling=0), dict(quality=55, subsampling=2, optimize=True), dict(quality=92, subsampling=1)]):
        p = tmp_path / f"prog{k}.jpg"
        Image.fromarray(pic).save(p, progressive=True, **kw)
        files.append(p); good.append(str(p))
    n_flip = len(good)
    s = tmp_path / "sniff.img"
    s.write_bytes(open(good[0], "rb").read())
    files.append(s); good.append(str(s))
    small = tmp_path / "small.jpg"
    Image.fromarray(pic[:9, :11]).save(small, quality=50, subsampling=2)
    base = small.read_bytes()
    for n in range(0, len(base)):                                                    # every truncation point
        q = tmp_path / f"trunc{n}.jpg"; q.write_bytes(base[:n]); files.append(q)
    psmall = tmp_path / "psmall.jpg"
    Image.fromarray(pic[:9, :11]).save(psmall, quality=50, subsampling=2, progressive=True)
    pbase = psmall.read_bytes()
    for n in range(0, len(pbase)):
        q = tmp_path / f"ptrunc{n}.jpg"; q.write_bytes(pbase[:n]); files.append(q)
    for j, gp in enumerate(good[:n_flip]):
        gb = open(gp, "rb").read()
        for k in range(100):
            b = bytearray(gb)
            for _ in range(int(rng.integers(1, 5))):
                b[int(rng.integers(2, len(b)))] = int(rng.integers(0, 256))
            q = tmp_path / f"flip{j}_{k}.jpg"; q.write_bytes(bytes(b)); files.append(q)
    soi = b"\xff\xd8"
    def seg(m, d):
        return bytes([0xff, m]) + struct.pack(">H", len(d) + 2) + d
    adversarial = {
        "huge.jpg": soi + seg(0xc0, struct.pack(">BHHB", 8, 65535, 65535, 3) + bytes([1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1])),
        "zero.jpg": soi + seg(0xc0, struct.pack(">BHHB", 8, 0, 16, 1) + bytes([1, 0x11, 0])),
        "nocomp.jpg": soi + seg(0xc0, struct.pack(">BHHB", 8, 16, 16, 0)),
        "badsamp.jpg": soi + seg(0xc0, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x00, 0])),
        "sos_first.jpg": soi + seg(0xda, bytes([1, 1, 0, 0, 63, 0])) + b"\x00" * 50,
        "notables.jpg": soi + seg(0xc0, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])) + seg(0xda, bytes([1, 1, 0, 0, 63, 0])) + b"\x12\x34" * 40 + b"\xff\xd9",
        "dht_overflow.jpg": soi + seg(0xc4, bytes([0x00] + [255] * 16) + b"\x00" * 100),
        "dht_kraft.jpg": soi + seg(0xc4, bytes([0x00, 3] + [0] * 15) + b"\x00\x01\x02"),
        "dqt_short.jpg": soi + seg(0xdb, bytes([0x00]) + b"\x01" * 10),
        "lenlie.jpg": soi + b"\xff\xe0\xff\xff" + b"\x00" * 30,
        "len1.jpg": soi + b"\xff\xe0\x00\x01" + b"\x00" * 30,
        "onlysoi.jpg": soi, "empty.jpg": b"",
        # progressive frame, scans with out-of-range bands / approximation bits / an AC scan over two components
        "prog_band.jpg": soi + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])) + seg(0xda, bytes([1, 1, 0, 5, 70, 0])) + b"\x12" * 40,
        "prog_al.jpg": soi + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])) + seg(0xda, bytes([1, 1, 0, 0, 0, 0x0f])) + b"\x12" * 40,
        "prog_ac2.jpg": soi + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 3) + bytes([1, 0x11, 0, 2, 0x11, 0, 3, 0x11, 0])) + seg(0xda, bytes([2, 1, 0, 2, 0, 1, 5, 0])) + b"\x12" * 40,
        "prog_refine_first.jpg": soi + seg(0xdb, bytes([0] + [1] * 64)) + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0]))
                                 + seg(0xda, bytes([1, 1, 0, 0, 0, 0x10])) + b"\xaa" * 40 + b"\xff\xd9",
    }
    # sampling factors the encoders above never produce (round-2 advisor finding: a 3:2 ratio took the 1:1 copy over a smaller plane).  Hand-built streams:
    # all-ones quantisation table, one-code Huffman tables (DC size 0 / AC end-of-block), so every block is the two bits "00" and zero bytes are valid data
    dqt = seg(0xdb, bytes([0] + [1] * 64))
    dht = seg(0xc4, bytes([0x00, 1] + [0] * 15 + [0])) + seg(0xc4, bytes([0x10, 1] + [0] * 15 + [0]))
    def frame(w, h, samp, prog):
        sof = seg(0xc2 if prog else 0xc0, struct.pack(">BHHB", 8, h, w, len(samp)) + b"".join(bytes([k + 1, (a << 4) | b, 0]) for k, (a, b) in enumerate(samp)))
        sos = seg(0xda, bytes([len(samp)] + [x for k in range(len(samp)) for x in (k + 1, 0)] + ([0, 0, 0] if prog else [0, 63, 0])))
        return soi + dqt + dht + sof + sos + b"\x00" * 400 + b"\xff\xd9"
    fractional, integral = {}, {}
    for prog in (False, True):
        t = "p" if prog else "s"
        fractional[f"frac33_{t}.jpg"] = frame(24, 24, [(3, 3), (2, 2), (2, 2)], prog)
        fractional[f"frac31_{t}.jpg"] = frame(24, 8, [(3, 1), (2, 1), (2, 1)], prog)
        fractional[f"frac13_{t}.jpg"] = frame(8, 24, [(1, 3), (1, 2), (1, 2)], prog)
        fractional[f"frac44_{t}.jpg"] = frame(32, 32, [(4, 4), (3, 3), (1, 1)], prog)
        fractional[f"frac_mixed_{t}.jpg"] = frame(29, 31, [(2, 3), (2, 2), (1, 1)], prog)
        integral[f"int44_{t}.jpg"] = frame(33, 35, [(4, 4), (1, 1), (2, 2)], prog)      # 4:1 replication beside a 2:1 triangle filter
        integral[f"int31_{t}.jpg"] = frame(25, 9, [(3, 1), (1, 1), (1, 1)], prog)
        integral[f"int14_{t}.jpg"] = frame(7, 37, [(1, 4), (1, 2), (1, 1)], prog)
        integral[f"int42_{t}.jpg"] = frame(17, 17, [(4, 2), (2, 2), (2, 1)], prog)
        integral[f"int_cmax_{t}.jpg"] = frame(19, 21, [(1, 1), (2, 2), (1, 2)], prog)   # luma is NOT the largest component
    adversarial.update(fractional)
    # scans are cheap to write and expensive to walk: 200 DC-refinement scans of 14 bytes each, and a scan with no entropy-coded byte at all
    adversarial["many_scans.jpg"] = (soi + dqt + dht + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])) + seg(0xda, bytes([1, 1, 0, 0, 0, 0x01])) + b"\x00" * 8
                                     + (seg(0xda, bytes([1, 1, 0, 0, 0, 0x10])) + b"\x00") * 200 + b"\xff\xd9")
    adversarial["empty_scan.jpg"] = soi + dqt + dht + seg(0xc2, struct.pack(">BHHB", 8, 16, 16, 1) + bytes([1, 0x11, 0])) + seg(0xda, bytes([1, 1, 0, 0, 0, 0])) + b"\xff\xd9"
    for name, data in list(adversarial.items()) + list(integral.items()):
        q = tmp_path / name; q.write_bytes(data); files.append(q)
    # the mutation fuzzer aims at the frame header's sampling bytes too (random flips almost never hit those three bytes)
    for j, gp in enumerate(good[:n_flip]):
        gb = open(gp, "rb").read()
        at = max(gb.find(b"\xff\xc0"), gb.find(b"\xff\xc2"))
        ncomp = gb[at + 9]
        for k in range(40):
            b = bytearray(gb)
            for c in range(ncomp):
                if rng.random() < 0.7:
                    b[at + 11 + 3 * c] = (int(rng.integers(1, 5)) << 4) | int(rng.integers(1, 5))
            q = tmp_path / f"samp{j}_{k}.jpg"; q.write_bytes(bytes(b)); files.append(q)
    rc = run(parsers, files)
    assert all(rc[gp] == 0 for gp in good), {gp: rc[gp] for gp in good}
    for name in adversarial:
        if name != "prog_refine_first.jpg":               # (DC refinement bits without a first scan decode to something; it must only be memory-safe)
            assert rc[str(tmp_path / name)] < 0, name
    assert all(rc[str(tmp_path / name)] == 0 for name in integral), {n: rc[str(tmp_path / n)] for n in integral}
    assert rc[str(tmp_path / "trunc0.jpg")] < 0 and rc[str(tmp_path / f"trunc{len(base) // 2}.jpg")] < 0
