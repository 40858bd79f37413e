"""Writes the .jpg textures next to this script from the .png ones (Pillow / libjpeg is only the ENCODER here: what the decoders have to agree on
is the decoded bytes, which tests/test_ref_kat6.py pins against stb_image built from the reference's own source).  The files are committed -- an
encoder of another version may choose other Huffman tables, so re-running this is followed by `make -C oracle -f Makefile.ref fixtures` for
ref_kat6.json and by tests/golden/make_golden.py for the ldr_texture3 image.

One file per decoder path:
  albedo420.jpg        4:2:0, standard Huffman tables            (2x2 chroma upsampling)
  albedo422.jpg        4:2:2, optimised tables, restart markers  (horizontal upsampling, DRI / RSTn)
  tiles444.jpg         4:4:4, 50 x 37                            (partial MCUs at the right and bottom edges)
  rough_grey.jpg       one component
  albedo_prog.jpg      progressive, 4:2:0                        (DC first/refine, AC first/refine scans of libjpeg's default script)
  tiles_prog444.jpg    progressive, 4:4:4, 50 x 37               (non-interleaved scans over partial blocks)
  rough_prog_grey.jpg  progressive, one component, restart markers inside the scans
"""
import os
from PIL import Image

here = os.path.dirname(os.path.abspath(__file__))
albedo = Image.open(os.path.join(here, "albedo.png")).convert("RGB")
rough = Image.open(os.path.join(here, "rough.png")).convert("L")
tiles = Image.open(os.path.join(here, "tiles.png")).convert("RGB").resize((50, 37))

jobs = [
    (albedo, "albedo420.jpg", dict(quality=92, subsampling=2, optimize=False)),
    (albedo, "albedo422.jpg", dict(quality=85, subsampling=1, optimize=True, restart_marker_blocks=3)),
    (tiles, "tiles444.jpg", dict(quality=95, subsampling=0)),
    (rough, "rough_grey.jpg", dict(quality=90)),
    (albedo, "albedo_prog.jpg", dict(quality=90, progressive=True)),
    (tiles, "tiles_prog444.jpg", dict(quality=88, subsampling=0, progressive=True)),
    (rough, "rough_prog_grey.jpg", dict(quality=80, progressive=True, restart_marker_blocks=5)),
]
for image, name, options in jobs:
    image.save(os.path.join(here, name), **options)
    print(name, os.path.getsize(os.path.join(here, name)))
