"""lk_load_image / lk_decode_image (include/lk_tracker.h, csrc/lk_image_io.cpp): the containers the reference's frames come in,
decoded to the 8-bit grey that cv::imread(path, IMREAD_GRAYSCALE) hands managerClass (manager_class.cpp:102-107,174,211,250).
CPU only.  Two sources of files: PIL's encoders (a real-world encoder's filter choices, palettes, BMP layouts) and a small
writer of its own below (every PNG colour type x bit depth, every filter type, Adam7).  Grey content must come back exactly;
colour goes through the two fixed-point conversions restated here."""
import io
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from correlation_amd import tracker as tk

PIL = pytest.importorskip("PIL.Image")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def grey_libpng(rgb):   # png_set_rgb_to_gray(0.299, 0.587): 15-bit coefficients
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((9797 * r + 19234 * g + 3737 * b + 16384) >> 15).astype(np.uint8)


def grey_opencv(rgb):   # OpenCV's BGR -> grey of its BMP / PxM decoders: 14-bit coefficients
    r, g, b = (rgb[..., i].astype(np.uint32) for i in range(3))
    return ((4899 * r + 9617 * g + 1868 * b + 8192) >> 14).astype(np.uint8)


def png_bytes(samples, depth, ctype, palette=None, interlace=False):
    """samples: (h, w, channels) integers of `depth` bits; filter types cycle 0..4 over the rows of every pass"""
    h, w, ch = samples.shape

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data))

    def pack_row(row):   # (w, ch) -> bytes
        if depth == 16:
            return row.astype(">u2").tobytes()
        if depth == 8:
            return row.astype(np.uint8).tobytes()
        bits = "".join(format(int(v), f"0{depth}b") for v in row.reshape(-1))
        bits += "0" * (-len(bits) % 8)
        return bytes(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))

    def filtered(rows, bpp):
        out, prev = bytearray(), bytes(len(rows[0])) if rows else b""
        for k, cur in enumerate(rows):
            f = k % 5
            line = bytearray(len(cur))
            for i, x in enumerate(cur):
                a = cur[i - bpp] if i >= bpp else 0
                b = prev[i]
                c = prev[i - bpp] if i >= bpp else 0
                if f == 4:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if pa <= pb and pa <= pc else (b if pb <= pc else c)
                else:
                    pred = (0, a, b, (a + b) >> 1)[f]
                line[i] = (x - pred) & 255
            out += bytes([f]) + line
            prev = cur
        return bytes(out)

    bpp = max(1, depth * ch // 8)
    if interlace:
        passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    else:
        passes = [(0, 0, 1, 1)]
    raw = b""
    for x0, y0, dx, dy in passes:
        sub = samples[y0::dy, x0::dx]
        if sub.shape[0] and sub.shape[1]:
            raw += filtered([pack_row(r) for r in sub], bpp)
    z = zlib.compress(raw, 6)
    idat = chunk(b"IDAT", z[:len(z) // 2]) + chunk(b"IDAT", z[len(z) // 2:])   # (split on purpose)
    out = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 1 if interlace else 0))
    out += chunk(b"gAMA", struct.pack(">I", 45455))   # an ancillary chunk to skip
    if palette is not None:
        out += chunk(b"PLTE", palette.astype(np.uint8).tobytes())
    return out + idat + chunk(b"IEND", b"")


def corpus():
    rng = np.random.default_rng(5)
    files = {}

    def pil(name, img, fmt, expect, **kw):
        buf = io.BytesIO()
        img.save(buf, fmt, **kw)
        files[name] = (buf.getvalue(), expect)

    g8 = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    g8[5:20, 7:30] = np.arange(23, dtype=np.uint8)[None, :] * 9      # (smooth parts: the encoder picks other filters there)
    rgb = rng.integers(0, 256, (29, 31, 3), dtype=np.uint8)
    rgb[3:9, 4:20] = 77                                               # r == g == b pixels
    pil("pil_grey8.png", PIL.fromarray(g8), "PNG", g8)
    pil("pil_grey8_optimized.png", PIL.fromarray(g8), "PNG", g8, optimize=True)
    g16 = rng.integers(0, 65536, (19, 23), dtype=np.uint16)
    pil("pil_grey16.png", PIL.fromarray(g16), "PNG", (g16 >> 8).astype(np.uint8))
    pil("pil_rgb.png", PIL.fromarray(rgb), "PNG", grey_libpng(rgb))
    rgba = np.dstack([rgb, rng.integers(0, 256, rgb.shape[:2], dtype=np.uint8)])
    pil("pil_rgba.png", PIL.fromarray(rgba), "PNG", grey_libpng(rgb))
    la = np.dstack([g8, 255 - g8])
    pil("pil_grey_alpha.png", PIL.fromarray(la), "PNG", g8)
    bw = rng.integers(0, 2, (17, 21), dtype=np.uint8)
    pil("pil_1bit.png", PIL.fromarray(bw * 255).convert("1"), "PNG", bw * 255)
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    idx = rng.integers(0, 256, (23, 19), dtype=np.uint8)
    p = PIL.fromarray(idx)
    p.putpalette(pal.reshape(-1).tolist())
    pil("pil_palette.png", p, "PNG", grey_libpng(pal[idx]))
    pil("pil_grey8.bmp", PIL.fromarray(g8), "BMP", g8)               # 8-bit with a grey palette, rows padded to 4 bytes
    pil("pil_rgb.bmp", PIL.fromarray(rgb), "BMP", grey_opencv(rgb))
    pil("pil_palette.bmp", p, "BMP", grey_opencv(pal[idx]))
    pil("pil_1bit.bmp", PIL.fromarray(bw * 255).convert("1"), "BMP", bw * 255)
    pil("pil_grey8.pgm", PIL.fromarray(g8), "PPM", g8)
    pil("pil_rgb.ppm", PIL.fromarray(rgb), "PPM", grey_opencv(rgb))
    # the writer above: every colour type x depth, all five filters, with and without Adam7 (sizes that leave passes empty too)
    for h, w in ((11, 13), (1, 1), (2, 3), (9, 4)):
        for interlace in (False, True):
            tag = f"{h}x{w}{'_adam7' if interlace else ''}"
            for depth in (1, 2, 4, 8, 16):
                s = rng.integers(0, 1 << depth, (h, w, 1))
                expect = (s[..., 0] >> 8 if depth == 16 else s[..., 0] * (255 // ((1 << depth) - 1))).astype(np.uint8)
                files[f"own_grey{depth}_{tag}.png"] = (png_bytes(s, depth, 0, interlace=interlace), expect)
            for depth in (1, 2, 4, 8):
                n = 1 << depth
                palette = rng.integers(0, 256, (n, 3))
                s = rng.integers(0, n, (h, w, 1))
                files[f"own_palette{depth}_{tag}.png"] = (png_bytes(s, depth, 3, palette, interlace), grey_libpng(palette[s[..., 0]]))
            for depth in (8, 16):
                hi = (lambda v: (v >> 8 if depth == 16 else v))
                s = rng.integers(0, 1 << depth, (h, w, 4))
                files[f"own_rgb{depth}_{tag}.png"] = (png_bytes(s[..., :3], depth, 2, interlace=interlace), grey_libpng(hi(s[..., :3])))
                files[f"own_rgba{depth}_{tag}.png"] = (png_bytes(s, depth, 6, interlace=interlace), grey_libpng(hi(s[..., :3])))
                files[f"own_ga{depth}_{tag}.png"] = (png_bytes(s[..., :2], depth, 4, interlace=interlace), hi(s[..., 0]).astype(np.uint8))
    # PNM by hand: ascii with comments, 16-bit binary, bitmaps
    small = rng.integers(0, 256, (3, 5), dtype=np.uint8)
    files["ascii.pgm"] = (b"P2\n# a comment\n5 3\n# another\n255\n" + " ".join(str(v) for v in small.reshape(-1)).encode() + b"\n", small)
    c3 = rng.integers(0, 256, (3, 5, 3), dtype=np.uint8)
    files["ascii.ppm"] = (b"P3 5 3 255\n" + "\n".join(str(v) for v in c3.reshape(-1)).encode(), grey_opencv(c3))
    w16 = rng.integers(0, 65536, (4, 6), dtype=np.uint16)
    files["wide.pgm"] = (b"P5\n6 4\n65535\n" + w16.astype(">u2").tobytes(), (w16 >> 8).astype(np.uint8))
    bits = rng.integers(0, 2, (5, 11), dtype=np.uint8)
    files["bitmap.pbm"] = (b"P4\n11 5\n" + np.packbits(bits, axis=1).tobytes(), ((1 - bits) * 255).astype(np.uint8))
    files["ascii.pbm"] = (b"P1\n# c\n11 5\n" + "".join(str(v) for v in bits.reshape(-1)).encode(), ((1 - bits) * 255).astype(np.uint8))
    # TIFF: libtiff's encoders through PIL (strips; every compression, with and without horizontal differencing) ...
    g8b = np.ascontiguousarray(g8[:, :50])
    for comp in (None, "packbits", "tiff_lzw", "tiff_adobe_deflate"):
        pil(f"pil_grey8_{comp}.tif", PIL.fromarray(g8b), "TIFF", g8b, compression=comp)
    for comp in ("tiff_lzw", "tiff_adobe_deflate"):
        pil(f"pil_grey8_{comp}_pred.tif", PIL.fromarray(g8b), "TIFF", g8b, compression=comp, tiffinfo={317: 2})
        pil(f"pil_grey16_{comp}_pred.tif", PIL.fromarray(g16), "TIFF", (g16 >> 8).astype(np.uint8), compression=comp, tiffinfo={317: 2})
        pil(f"pil_rgb_{comp}_pred.tif", PIL.fromarray(rgb), "TIFF", grey_opencv(rgb), compression=comp, tiffinfo={317: 2})
    smooth = (np.add.outer(np.arange(300), np.arange(200)) // 3 % 256).astype(np.uint8)   # (long LZW strings, several strips)
    pil("pil_smooth_lzw.tif", PIL.fromarray(smooth), "TIFF", smooth, compression="tiff_lzw")
    noise = rng.integers(0, 256, (120, 90), dtype=np.uint8)                               # (the LZW table fills up and is cleared)
    pil("pil_noise_lzw.tif", PIL.fromarray(noise), "TIFF", noise, compression="tiff_lzw")
    pil("pil_grey16.tif", PIL.fromarray(g16), "TIFF", (g16 >> 8).astype(np.uint8))
    pil("pil_rgb.tif", PIL.fromarray(rgb), "TIFF", grey_opencv(rgb))
    pil("pil_rgba_lzw.tif", PIL.fromarray(rgba), "TIFF", grey_opencv(rgb), compression="tiff_lzw")
    pil("pil_palette.tif", p, "TIFF", grey_opencv(pal[idx]))
    pil("pil_grey_alpha.tif", PIL.fromarray(la), "TIFF", g8)

    # ... and by hand: big endian, tiles, white-is-zero, 16-bit differencing in both byte orders
    def tiff_bytes(samples, bps, photometric, big_endian=False, tile=None, predictor=1):
        h, w, ch = samples.shape
        e = ">" if big_endian else "<"
        dt = np.dtype(f"{e}u2") if bps == 16 else np.dtype(np.uint8)

        def encode(block):   # (rows, cols, ch) -> bytes, with the differencing applied
            b = block.astype(np.int64)
            if predictor == 2:
                b[:, 1:] = (b[:, 1:] - block[:, :-1]) % (1 << bps)
            return b.astype(dt).tobytes()

        if tile:
            th, tw = tile
            chunks = []
            for y in range(0, h, th):
                for x in range(0, w, tw):
                    blk = np.zeros((th, tw, ch), np.int64)
                    part = samples[y:y + th, x:x + tw]
                    blk[:part.shape[0], :part.shape[1]] = part
                    chunks.append(encode(blk))
        else:
            rps = 3
            chunks = [encode(samples[y:y + rps]) for y in range(0, h, rps)]
        entries = {256: (3, [w]), 257: (3, [h]), 258: (3, [bps] * ch), 259: (3, [1]), 262: (3, [photometric]), 277: (3, [ch]), 317: (3, [predictor])}
        n = len(chunks)
        data_at = 8
        offsets, at = [], data_at
        for c in chunks:
            offsets.append(at)
            at += len(c)
        if tile:
            entries.update({322: (3, [tile[1]]), 323: (3, [tile[0]]), 324: (4, offsets), 325: (4, [len(c) for c in chunks])})
        else:
            entries.update({278: (3, [3]), 273: (4, offsets), 279: (4, [len(c) for c in chunks])})
        ifd_at = at + (at & 1)
        body = b"".join(chunks) + (b"\0" if at & 1 else b"")
        extra_at = ifd_at + 2 + 12 * len(entries) + 4
        ifd, extra = struct.pack(e + "H", len(entries)), b""
        for tag in sorted(entries):
            typ, vals = entries[tag]
            fmt = "H" if typ == 3 else "I"
            raw = struct.pack(e + fmt * len(vals), *vals)
            if len(raw) <= 4:
                field = raw + bytes(4 - len(raw))
            else:
                field = struct.pack(e + "I", extra_at + len(extra))
                extra += raw + bytes(len(raw) & 1)
            ifd += struct.pack(e + "HHI", tag, typ, len(vals)) + field
        ifd += struct.pack(e + "I", 0)
        return (b"MM\0*" if big_endian else b"II*\0") + struct.pack(e + "I", ifd_at) + body + ifd + extra

    for big in (False, True):
        tag = "be" if big else "le"
        s8 = rng.integers(0, 256, (10, 13, 1))
        s16 = rng.integers(0, 65536, (10, 13, 1))
        c16 = rng.integers(0, 65536, (7, 9, 3))
        files[f"own_grey8_{tag}.tif"] = (tiff_bytes(s8, 8, 1, big), s8[..., 0].astype(np.uint8))
        files[f"own_grey8_white_is_zero_{tag}.tif"] = (tiff_bytes(s8, 8, 0, big), (255 - s8[..., 0]).astype(np.uint8))
        files[f"own_grey16_{tag}.tif"] = (tiff_bytes(s16, 16, 1, big), (s16[..., 0] >> 8).astype(np.uint8))
        files[f"own_grey16_pred_{tag}.tif"] = (tiff_bytes(s16, 16, 1, big, predictor=2), (s16[..., 0] >> 8).astype(np.uint8))
        files[f"own_rgb16_pred_{tag}.tif"] = (tiff_bytes(c16, 16, 2, big, predictor=2), grey_opencv(c16 >> 8))
        files[f"own_grey8_tiles_{tag}.tif"] = (tiff_bytes(s8, 8, 1, big, tile=(16, 16)), s8[..., 0].astype(np.uint8))
        files[f"own_grey16_tiles_pred_{tag}.tif"] = (tiff_bytes(s16, 16, 1, big, tile=(16, 16), predictor=2), (s16[..., 0] >> 8).astype(np.uint8))
        wide = rng.integers(0, 256, (40, 37, 1))
        files[f"own_grey8_many_tiles_{tag}.tif"] = (tiff_bytes(wide, 8, 1, big, tile=(16, 16)), wide[..., 0].astype(np.uint8))
    # BMP by hand: top-down 32-bit and a 4-bit palette image
    bgra = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    hdr = struct.pack("<IiiHHIIiiII", 40, 7, -5, 1, 32, 0, 0, 0, 0, 0, 0)
    files["topdown32.bmp"] = (b"BM" + struct.pack("<IHHI", 14 + 40 + bgra.size, 0, 0, 54) + hdr + bgra.tobytes(), grey_opencv(bgra[..., 2::-1]))
    pal4 = rng.integers(0, 256, (16, 4), dtype=np.uint8)
    idx4 = rng.integers(0, 16, (3, 5), dtype=np.uint8)
    rows4 = b"".join(bytes([(r[0] << 4) | r[1], (r[2] << 4) | r[3], r[4] << 4, 0]) for r in idx4[::-1])
    hdr = struct.pack("<IiiHHIIiiII", 40, 5, 3, 1, 4, 0, 0, 0, 0, 16, 0)
    files["palette4.bmp"] = (b"BM" + struct.pack("<IHHI", 14 + 40 + 64 + len(rows4), 0, 0, 118) + hdr + pal4.tobytes() + rows4, grey_opencv(pal4[idx4][..., 2::-1]))
    return files


@pytest.fixture(scope="module")
def files():
    return corpus()


def test_every_container_decodes_to_the_expected_grey(engine_lib, files):
    assert len(files) > 150
    for name, (data, expect) in files.items():
        got = tk.decode_image(data, engine_lib)
        assert got.dtype == np.uint8 and got.shape == expect.shape, name
        assert np.array_equal(got, expect), name


def test_grey_files_equal_pils_own_decoders(engine_lib, files):
    for name in ("pil_grey8.png", "pil_grey8_optimized.png", "pil_grey_alpha.png", "pil_1bit.png", "pil_grey8.bmp", "pil_grey8.pgm", "pil_1bit.bmp",
                 "pil_grey8_None.tif", "pil_grey8_packbits.tif", "pil_grey8_tiff_lzw.tif", "pil_grey8_tiff_adobe_deflate_pred.tif",
                 "pil_smooth_lzw.tif", "pil_noise_lzw.tif", "own_grey8_tiles_be.tif", "own_grey8_many_tiles_le.tif"):
        ours = tk.decode_image(files[name][0], engine_lib)
        theirs = np.asarray(PIL.open(io.BytesIO(files[name][0])).convert("L"))
        assert np.array_equal(ours, theirs), name
    # colour: PIL's "L" is the ITU-R 601 formula in another fixed point - within one grey level of both conversions here
    for name in ("pil_rgb.png", "pil_rgb.bmp", "pil_rgb.ppm", "pil_palette.png"):
        ours = tk.decode_image(files[name][0], engine_lib).astype(int)
        theirs = np.asarray(PIL.open(io.BytesIO(files[name][0])).convert("RGB").convert("L")).astype(int)
        assert np.abs(ours - theirs).max() <= 1, name


def test_load_image_reads_files_and_the_old_pgm_reader_agrees(engine_lib, files, tmp_path):
    for name in ("pil_grey8.png", "pil_rgb.bmp", "wide.pgm", "pil_grey8.pgm"):
        path = tmp_path / name
        path.write_bytes(files[name][0])
        assert np.array_equal(tk.load_image(path, engine_lib), files[name][1]), name
    assert np.array_equal(tk.load_pgm(str(tmp_path / "pil_grey8.pgm"), engine_lib), files["pil_grey8.pgm"][1])


def test_malformed_files_are_refused_not_guessed(engine_lib, files, tmp_path):
    png = files["pil_grey8.png"][0]
    bad = {
        "missing file": None,
        "empty": b"",
        "jpeg": b"\xff\xd8\xff\xe0" + bytes(64),
        "tiff without an image": b"II*\x00" + bytes(64),
        "bigtiff": b"II+\x00" + bytes(64),
        "tiff with float samples": files["own_grey16_le.tif"][0].replace(struct.pack("<HHIHH", 317, 3, 1, 1, 0), struct.pack("<HHIHH", 339, 3, 1, 3, 0)),
        "truncated png": png[:len(png) // 2],
        "png without IEND": png[:-12],
        "bad crc": png[:40] + bytes([png[40] ^ 1]) + png[41:],
        "huge header, tiny data": png_bytes(np.zeros((2, 2, 1), int), 8, 0)[:16] + struct.pack(">II", 1 << 15, 1 << 15) + b"\x08\x00\x00\x00\x00" + bytes(64),
        "short pgm": b"P5\n4 4\n255\n" + bytes(15),
        "pgm without maxval": b"P5\n4 4\n",
        "rle bmp": files["pil_grey8.bmp"][0][:30] + struct.pack("<I", 1) + files["pil_grey8.bmp"][0][34:],
        "truncated bmp": files["pil_rgb.bmp"][0][:200],
        "palette index out of range": png_bytes(np.full((2, 2, 1), 3), 2, 3, np.zeros((2, 3), int)),
    }
    for what, data in bad.items():
        if data is None:
            with pytest.raises(IOError):
                tk.load_image(tmp_path / "nope.png", engine_lib)
            continue
        with pytest.raises(IOError):
            tk.decode_image(data, engine_lib)
        (tmp_path / "bad.bin").write_bytes(data)
        with pytest.raises(IOError):
            tk.load_image(tmp_path / "bad.bin", engine_lib)


def test_decoders_under_asan_and_ubsan_on_damaged_files(files, tmp_path):
    """every file of the corpus, then truncated and byte-flipped copies of it (PNG chunk CRCs repaired, so that the damage
    reaches the inflater, the filters and the pixel loops): whatever the outcome, no read or write outside a buffer"""
    exe = tmp_path / "image_io_driver"
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                        "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "include"),
                        os.path.join(ROOT, "tests", "host", "image_io_driver.cpp"),
                        os.path.join(ROOT, "correlation_amd", "csrc", "lk_image_io.cpp"), "-lz", "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    corpus_dir = tmp_path / "corpus"
    corpus_dir.mkdir()
    for name, (data, expect) in files.items():
        (corpus_dir / name).write_bytes(data)
        (corpus_dir / (name + ".expect")).write_bytes(struct.pack("<ii", *expect.shape) + expect.tobytes())
    r = subprocess.run([str(exe), str(corpus_dir)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert r.returncode == 0 and "image_io_driver ok" in r.stdout, r.stdout[-1000:] + r.stderr[-6000:]
