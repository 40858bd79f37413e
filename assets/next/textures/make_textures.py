#!/usr/bin/env python3
"""Writes the three small 8-bit PNG textures of assets/next/ldr_texture.json (deterministic; committed next to this script).
They exist to exercise the LDR texel path: RGB (colour type 2), RGBA (6) and palette (3) files, every scanline filter type,
zlib level 9 (dynamic Huffman blocks), all 256 byte values in the RGB file (the gamma table)."""
import os
import struct
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))


def chunk(tag, data):
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)


def write_png(path, w, h, color_type, rows, palette=None, trns=None):
    bpp = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[color_type]
    raw = bytearray()
    prev = bytes(w*bpp)
    for y, row in enumerate(rows):
        f = y % 5  # cycle through None, Sub, Up, Average, Paeth
        out = bytearray()
        for x in range(w*bpp):
            a = row[x - bpp] if x >= bpp else 0
            b = prev[x]
            c = prev[x - bpp] if x >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][f]
            out.append((row[x] - pred) & 0xFF)
        raw.append(f)
        raw += out
        prev = bytes(row)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color_type, 0, 0, 0))
    if palette:
        png += chunk(b"PLTE", bytes(palette))
    if trns:
        png += chunk(b"tRNS", bytes(trns))
    z = zlib.compress(bytes(raw), 9)
    half = len(z)//2  # two IDAT chunks: the stream must be joined before inflating
    png += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")
    open(path, "wb").write(png)


def lcg(seed):
    s = seed
    while True:
        s = (s*1664525 + 1013904223) & 0xFFFFFFFF
        yield s >> 24


def main():
    # albedo.png: 64x48 RGB; the first row holds bytes 0..63, ..., so that R runs over all 256 values; G, B are patterns + noise
    w, h = 64, 48
    rnd = lcg(1)
    rows = []
    for y in range(h):
        row = bytearray()
        for x in range(w):
            r = (y*w + x) & 0xFF
            g = (x*4 + (next(rnd) & 15)) & 0xFF
            b = 255 - ((y*5 + (x ^ y)) & 0xFF)
            row += bytes((r, g, b))
        rows.append(row)
    write_png(os.path.join(HERE, "albedo.png"), w, h, 2, rows)

    # rough.png: 32x32 RGBA, a smooth roughness bump (used through REQUEST_AVERAGE: integer (r + g + b)/3, no gamma)
    w = h = 32
    rows = []
    for y in range(h):
        row = bytearray()
        for x in range(w):
            d = ((x - 16)**2 + (y - 16)**2)**0.5
            v = max(8, min(200, int(20 + 9*d)))
            row += bytes((v, (v + 1) & 0xFF, (v + 3) & 0xFF, 128 + (x & 63)))
        rows.append(row)
    write_png(os.path.join(HERE, "rough.png"), w, h, 6, rows)

    # tiles.png: 16x16 palette image (8 colours)
    pal = [200, 40, 30, 30, 160, 60, 40, 60, 190, 220, 200, 40, 230, 230, 230, 20, 20, 20, 150, 60, 170, 60, 170, 170]
    w = h = 16
    rows = [bytearray(((x//2 + 3*(y//2)) % 8) for x in range(w)) for y in range(h)]
    write_png(os.path.join(HERE, "tiles.png"), w, h, 3, rows, palette=pal)


if __name__ == "__main__":
    main()
