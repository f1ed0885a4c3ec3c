#!/usr/bin/env python3
"""Fixture from the reference's OWN published artefacts: illusions_rating/EIGEN-images/<name>/{small,vectors}.png.

`vectors.png` is the flow overlay lucas_kanade() saved for a stimulus (yellow radius-2 dots at the tracked
goodFeaturesToTrack corners, red flow lines), `small.png` the 160x120 stimulus.  Stored: the stimulus pixels and the
mask of yellow overlay pixels -- data only.  tests/test_oracle_golden.py uses them as EVIDENCE for the recalled
goodFeaturesToTrack parameters (the overlay was drawn on the PredNet prediction of the stimulus, not on the stimulus
itself, so corner positions agree only partly).
Run in the build container only:  python tests/golden/make_flow_evidence.py
"""
import os

import numpy as np
from PIL import Image

BASE = "/root/reference/illusions_rating/EIGEN-images/"
OUT = os.path.dirname(os.path.abspath(__file__))
CASES = [("expand_01", "small.png", "vectors.png"), ("expand_02", "small.png", "vectors.png"), ("rotate_02", "small.png", "vectors.png"),
         ("manyfish", "manyfish-small.png", "manyfish-vectors.png"), ("color_01_expand", "small.png", "vectors.png")]

out = {}
for d, small, vec in CASES:
    im = np.asarray(Image.open(BASE + d + "/" + small))
    v = np.asarray(Image.open(BASE + d + "/" + vec).convert("RGB")).astype(int)
    out[d + "_img"] = np.ascontiguousarray(im.transpose(2, 0, 1) if im.ndim == 3 else im[None]).astype(np.uint8)
    out[d + "_yellow"] = np.packbits((v[:, :, 0] > 200) & (v[:, :, 1] > 200) & (v[:, :, 2] < 90))
np.savez_compressed(os.path.join(OUT, "flow_overlays.npz"), **out)
print("written", os.path.join(OUT, "flow_overlays.npz"))
