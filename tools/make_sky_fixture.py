#!/usr/bin/env python3
"""Writes the procedural environment of the tea-set configuration (scenes.sky_map: sun disk + gradient, the stand-in
for the HDR file the reference's scene names but does not ship) as Radiance .hdr fixtures under tests/golden/:
sky_512x256_flat.hdr (flat RGBE pixels) and sky_512x256_rle.hdr (run-length encoded scanlines), both written by the
C++ host's own writer (host/image_io.hpp through libhostio.so)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
sky = pkg.scenes.sky_map(512, 256)
for rle in (False, True):
    path = os.path.join(ROOT, "tests", "golden", "sky_512x256_%s.hdr" % ("rle" if rle else "flat"))
    pkg.hostio.save_hdr(path, sky, rle=rle)
    print(path, os.path.getsize(path))
