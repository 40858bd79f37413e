#!/usr/bin/env python3
"""Writes the .bmp / .tga twins of the three PNG textures (assets/next/ldr_texture2.json): the formats the reference decodes through stb_image
(io/ImageIO.cpp:489-490).  Written by hand, byte by byte, so that each file exercises one form of its format:
  albedo.bmp  24 bits, bottom-up, rows padded to 4 bytes          tiles.bmp   8-bit palette, TOP-DOWN (negative height)
  rough.tga   8-bit grey, run-length encoded, bottom-up           albedo.tga  32 bits (B, G, R, A), raw, top-down
Needs pillow only to read the PNGs."""
import os
import struct
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))


def write_bmp24(path, img):
    w, h = img.size
    px = img.convert("RGB").load()
    row = (w*3 + 3)//4*4
    body = bytearray()
    for y in range(h - 1, -1, -1):
        line = bytearray()
        for x in range(w):
            r, g, b = px[x, y]
            line += bytes((b, g, r))
        body += line + bytes(row - len(line))
    open(path, "wb").write(b"BM" + struct.pack("<IHHI", 54 + len(body), 0, 0, 54) + struct.pack("<IiiHHIIiiII", 40, w, h, 1, 24, 0, len(body), 2835, 2835, 0, 0) + body)


def write_bmp8_topdown(path, img):
    pal_img = img.convert("RGB").quantize(256) if img.mode != "P" else img
    w, h = pal_img.size
    pal = pal_img.getpalette()[:768]
    pal += [0]*(768 - len(pal))
    table = b"".join(bytes((pal[i*3 + 2], pal[i*3 + 1], pal[i*3], 0)) for i in range(256))
    row = (w + 3)//4*4
    idx = pal_img.load()
    body = bytearray()
    for y in range(h):
        body += bytes(idx[x, y] for x in range(w)) + bytes(row - w)
    off = 54 + 1024
    open(path, "wb").write(b"BM" + struct.pack("<IHHI", off + len(body), 0, 0, off) + struct.pack("<IiiHHIIiiII", 40, w, -h, 1, 8, 0, len(body), 2835, 2835, 256, 0) + table + body)


def write_tga_grey_rle(path, img):
    g = img.convert("L")
    w, h = g.size
    px = g.load()
    body = bytearray()
    for y in range(h - 1, -1, -1):                                   # bottom-up (descriptor bit 5 clear)
        x = 0
        while x < w:
            run = 1
            while x + run < w and run < 128 and px[x + run, y] == px[x, y]:
                run += 1
            if run > 1:
                body += bytes((0x80 | (run - 1), px[x, y]))
                x += run
            else:
                lit = 1
                while x + lit < w and lit < 128 and (x + lit + 1 >= w or px[x + lit, y] != px[x + lit + 1, y]):
                    lit += 1
                body += bytes((lit - 1,)) + bytes(px[x + k, y] for k in range(lit))
                x += lit
    open(path, "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 11, 0, 0, 0, 0, 0, w, h, 8, 0) + body)


def write_tga32_topdown(path, img):
    rgba = img.convert("RGBA")
    w, h = rgba.size
    px = rgba.load()
    body = bytearray()
    for y in range(h):
        for x in range(w):
            r, g, b, a = px[x, y]
            body += bytes((b, g, r, a))
    open(path, "wb").write(struct.pack("<BBBHHBHHHHBB", 0, 0, 2, 0, 0, 0, 0, 0, w, h, 32, 0x28) + body)


def main():
    albedo = Image.open(os.path.join(HERE, "albedo.png"))
    rough = Image.open(os.path.join(HERE, "rough.png"))
    tiles = Image.open(os.path.join(HERE, "tiles.png"))
    write_bmp24(os.path.join(HERE, "albedo.bmp"), albedo)
    write_bmp8_topdown(os.path.join(HERE, "tiles.bmp"), tiles)
    write_tga_grey_rle(os.path.join(HERE, "rough.tga"), rough)
    write_tga32_topdown(os.path.join(HERE, "albedo.tga"), albedo)
    for f in ("albedo.bmp", "tiles.bmp", "rough.tga", "albedo.tga"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
