"""The host loader's own .bmp / .tga / .jpg decoders against stb_image -- the decoder the reference hands these files to (io/ImageIO.cpp:489-490),
compiled from the reference's own thirdparty/stbi/stb_image.c by the committed recipe (oracle/Makefile.ref, driver oracle/ref_driver/ref_kat6.cpp):
width, height and EVERY RGBA byte of every file under assets/next/textures.  A JPEG's pixels depend on the decoder's arithmetic (IDCT, chroma
upsampling, colour conversion), so "close" is not the bar: identical is."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, REPO

KAT = json.load(open(os.path.join(GOLDEN, "ref_kat6.json")))
TEX = os.path.join(REPO, "assets", "next", "textures")


def decode(ta, name):
    lib = ta.host_lib()
    lib.tgh_decode_ldr.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_uint64, C.c_char_p, C.c_int]
    w, h = C.c_int(), C.c_int()
    buf = np.zeros(1 << 20, np.uint8)
    err = C.create_string_buffer(256)
    rc = lib.tgh_decode_ldr(os.path.join(TEX, name).encode(), C.byref(w), C.byref(h), buf.ctypes.data, buf.size, err, 256)
    if rc != 0:
        raise RuntimeError(err.value.decode())
    return w.value, h.value, buf[:w.value*h.value*4].reshape(h.value, w.value, 4)


@pytest.mark.parametrize("name", list(KAT))
def test_decoder_reproduces_stb_image(ta, name):
    want = KAT[name]
    w, h, got = decode(ta, name)
    assert (w, h) == (want["w"], want["h"])
    ref = np.frombuffer(bytes.fromhex(want["rgba"]), np.uint8).reshape(h, w, 4)
    rgb_same = (got[..., :3] == ref[..., :3])
    assert rgb_same.all(), f"{name}: {int((~rgb_same.all(axis=2)).sum())} of {w*h} pixels differ from stb_image (max {int(np.abs(got[..., :3].astype(int) - ref[..., :3].astype(int)).max())} levels)"
    assert np.array_equal(got[..., 3], ref[..., 3]), f"{name}: alpha differs"
